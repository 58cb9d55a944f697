import importlib, sys, json
sys.path.insert(0, '.')
pkg = importlib.import_module("bls-verify-gadget_amd")
# blocks = waves: 1024 = one wave per SIMD, 8192 = eight
for blocks in (1024, 2048, 8192):
    r = {w: pkg.microbench(w, iters=512, blocks=blocks) for w in (1, 4, 2, 3)}
    print(blocks, "fpmul28 %.3e  fpmul32 %.3e  ratio %.2f  inv %.3e (%.1f mul-times)  fp2mulw %.3e" % (r[1], r[4], r[1]/r[4], r[2], r[1]/r[2], r[3]))
