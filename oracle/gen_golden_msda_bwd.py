"""Golden vectors for the BACKWARD of the multi-scale deformable attention core op: the reference's own
`ms_deform_attn_core_pytorch` (OPS/functions/ms_deform_attn_func.py:55-77 -- the function its test checks the CUDA kernels against,
OPS/test.py:44-86) in float64 under torch.autograd, with a seeded upstream gradient.  Build container only.

    python oracle/gen_golden_msda_bwd.py      # writes tests/golden/g14_msda_bwd_*.npz
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import meta, save  # noqa: E402
from gen_golden_msda import core_inputs, load_ops  # noqa: E402


def main():
    fn, _ = load_ops()
    # first case = the reference's own gradient test shapes (OPS/test.py:24-28), then D = 32 / locations outside the map / D = 30
    for (N, M, D, Lq, shapes, P, seed, spread) in [(1, 2, 2, 2, [(6, 4), (3, 2)], 2, 3, 0.0),
                                                   (2, 8, 32, 50, [(16, 12), (8, 6), (4, 3)], 4, 31, 0.15),
                                                   (1, 4, 30, 33, [(9, 7), (5, 5)], 3, 32, 0.3),
                                                   (1, 8, 16, 40, [(7, 9)], 5, 33, 0.2)]:
        value, loc, aw = core_inputs(N, M, D, Lq, shapes, P, seed, scale=1.0, spread=spread)
        v, l, a = (t.double().requires_grad_(True) for t in (value, loc, aw))
        out = fn.ms_deform_attn_core_pytorch(v, torch.as_tensor(shapes, dtype=torch.long), l, a)
        g = torch.Generator().manual_seed(seed + 100)
        go = torch.randn(out.shape, generator=g)
        out.backward(go.double())
        save(f"g14_msda_bwd_N{N}_M{M}_D{D}_Lq{Lq}_L{len(shapes)}_P{P}",
             meta=meta(N=N, M=M, D=D, Lq=Lq, shapes=shapes, P=P, seed=seed, spread=spread, scale=1.0), grad_output=go,
             out=out.detach().float(), grad_value=v.grad.float(), grad_sampling_loc=l.grad.float(), grad_attn_weight=a.grad.float())


if __name__ == "__main__":
    main()
