"""Test-only helpers for the sharded map optimiser (CPU restatements injected in gloo tests)."""
import torch


def adam_reference(p, g, m, v, lr_col, step, eps, b1=0.9, b2=0.999):
    """torch.optim.Adam arithmetic with a per-column lr (test-only: injected in the gloo tests, checker of rtgs_fused_adam)."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p.sub_((lr_col[None, :] / bc1) * (m / denom))


