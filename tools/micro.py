import importlib, sys
sys.path.insert(0, '.')
p = importlib.import_module("bls-verify-gadget_amd")
for blocks in (16, 256, 1024, 2048, 4096, 16384):
    r = [p.microbench(w, iters=(64 if w == 2 else 512), blocks=blocks) for w in (1, 2, 3)]
    # per-wave time of one op = 64 lanes * blocks / rate / (waves running concurrently ~ blocks)
    print("waves=%6d  fp_mul/s %.3e  fp_inv/s %.3e  fp2-level products/s %.3e | fp_inv = %.1f fp_mul ; fp2-level eff = %.2f of fp_mul rate" % (
        blocks, r[0], r[1], r[2], r[0] / r[1], r[2] / r[0]))
