"""Error of the split-bf16 conv12 kernels against the exact-fp32 MFMA kernels, both measured against float64, on adversarial
operands.  GPU only:  python scripts/dev/split_error.py [n_img [case ...]]

The cases and the float64 reference live in tests/split_cases.py (shared with tests/test_gpu_parity.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "what-matters-for-meta-learning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mlhot                                    # noqa: E402
import split_cases as SC                        # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    only = sys.argv[2:]                                  # optional case names
    lib = mlhot.lib()
    print(f"{'case':28s} {'quantity':6s} {'fp32 max':>10s} {'split max':>10s} {'ratio':>6s} | {'fp32 rms':>10s} {'split rms':>10s} {'ratio':>6s}")
    for name in SC.CASES:
        if only and name not in only:
            continue
        res = SC.measure(lib, name, n)
        for q, (e32, es) in res.items():
            print(f"{name:28s} {q:6s} {e32[0]:10.3e} {es[0]:10.3e} {es[0] / max(e32[0], 1e-300):6.2f} | "
                  f"{e32[1]:10.3e} {es[1]:10.3e} {es[1] / max(e32[1], 1e-300):6.2f}")


if __name__ == "__main__":
    main()
