import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "what-matters-for-meta-learning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mlhot, split_cases as SC
lib = mlhot.lib()
n = 4
for variant in ("pairs_only", "free_only", "both"):
    cpu = list(SC.make("cancel_exact", n))
    if variant == "pairs_only": cpu[5][:, 40:] = 0
    if variant == "free_only": cpu[5][:, :40] = 0
    x, w1, b1, w2, b2, dp2 = [t.cuda() for t in cpu]
    lib.set_option("conv2_split", 0)
    p2, _, saved = lib.conv12_fwd(x, w1, b1, w2, b2)
    routes = lib.enc_routes(saved, n)
    _, ref = SC.ref64(*cpu, routes)
    for tag, bits in (("fp32", 0), ("split", 7)):
        lib.set_option("conv2_split", bits)
        got = lib.conv12_bwd(x, w1, b1, w2, dp2, saved)
        torch.cuda.synchronize()
        d1 = (got[0].double().cpu() - ref[0]); db = (got[1].double().cpu() - ref[1])
        print(variant, tag, "dw1 ref max %.3g err max %.3g mean %.3g | db1 ref max %.3g err max %.3g mean %.3g" % (
            ref[0].abs().max(), d1.abs().max(), d1.mean(), ref[1].abs().max(), db.abs().max(), db.mean()))
lib.set_option("conv2_split", 0)
