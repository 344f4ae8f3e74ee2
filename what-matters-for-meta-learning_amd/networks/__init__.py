"""Drop-in `networks` package: same module / class names, constructor signature,
forward signature and state_dict layout as the reference's networks/ (train.py:41-45),
with the arithmetic running in hand-written HIP kernels (libmlhot.so)."""
