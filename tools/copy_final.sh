#!/bin/bash
# gpurun_out/{final,pmcf,r06_seq_final} -> profiles/ (what tools/final_all.sh + tools/r06_seq.sh final left), run HERE after the call
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/pmcf/bench.json profiles/r06_bench.json
cp gpurun_out/final/kernel_table_headline.txt profiles/r06_headline_kernel_table.txt; cp gpurun_out/final/kernel_table_surface.txt profiles/r06_surface_kernel_table.txt
cp gpurun_out/final/kernel_stats_headline.csv profiles/r06_headline_kernel_stats.csv; cp gpurun_out/final/kernel_stats_surface.csv profiles/r06_surface_kernel_stats.csv
cp gpurun_out/pmcf/traffic_headline.json profiles/r06_traffic.json; cp gpurun_out/pmcf/traffic_surface.json profiles/r06_traffic_surface.json
cp gpurun_out/pmcf/traffic_headline.json profiles/traffic_latest.json; cp gpurun_out/pmcf/valu.json profiles/valu_latest.json; cp gpurun_out/pmcf/valu.json profiles/r06_valu.json
cp gpurun_out/pmcf/sq_summary_headline.txt profiles/r06_sq_summary_headline.txt; cp gpurun_out/pmcf/sq_summary_surface.txt profiles/r06_sq_summary_surface.txt
cp gpurun_out/pmcf/sq/pmc_sq_headline.csv profiles/r06_pmc_sq_headline.csv; cp gpurun_out/pmcf/sq/pmc_sq_surface.csv profiles/r06_pmc_sq_surface.csv
cp gpurun_out/r06_seq_final/table_sequence.txt profiles/r06_sequence_kernel_table.txt; cp gpurun_out/r06_seq_final/stage_profile.json profiles/r06_sequence_stage_profile.json; cp gpurun_out/r06_seq_final/seq.json profiles/r06_sequence_only.json
cp gpurun_out/final/pytest_gpu.txt profiles/r06_pytest_gpu.txt; cp gpurun_out/parity_margins.json profiles/r06_parity_margins.json
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
d = json.load(open('profiles/r06_bench.json'))
print(bench.source_hash(), d["roofline"]["source_sha16"], json.load(open('profiles/traffic_latest.json'))["_source_sha16"], json.load(open('profiles/valu_latest.json'))["_source_sha16"])
c = d["config"]; r = d["roofline"]
print("value", d["value"], d["ms_per_step"], "slam", d["slam_frames_per_sec"], "frac", r["frac"], r["frac_blend_fwd_plus_bwd"], r["frac_blend_fwd_plus_bwd_surface"],
      "raster", c["raster_fwd_bwd_ms"], c["raster_fwd_bwd_ms_surface"], "iter", c["map_iteration_ms"], c["map_iteration_ms_surface"], "icp", c["icp_track_ms"])
s = d["slam_sequence"]
print({k: s[k] for k in ("fps", "fps_tracking_plus_mapping", "mapping_ms_mean_optimised_frames", "mapping_ms_mean_other_frames", "tracking_ms_mean", "peak_device_memory_MB", "gaussians", "ate_rmse_m", "keyframes")}, s.get("plain_renders"))
s2 = json.load(open('profiles/r06_sequence_only.json'))["sequence"]
print("sequence only", {k: s2[k] for k in ("fps", "mapping_ms_mean_optimised_frames", "mapping_ms_mean_other_frames", "peak_device_memory_MB")})
PY
head -3 profiles/r06_headline_kernel_table.txt | cut -c1-90; head -4 profiles/r06_surface_kernel_table.txt | tail -3 | cut -c1-90
