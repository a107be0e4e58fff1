import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import common, or_cdchomp_amd
for dev in ("1", "0"):
    os.environ["ORC_SDF_DEVICE"] = dev
    mod = or_cdchomp_amd.Module(0)
    t0 = time.perf_counter()
    model = common.setup_product_tree30(mod)
    t1 = time.perf_counter()
    print("ORC_SDF_DEVICE=%s: config 5 scene (robot + 4 fields at cube_extent 0.005): %.2f s" % (dev, t1 - t0))
    for nm in ("box0", "box1"):
        try:
            sdf, lengths, pose = mod.get_sdf(nm); print("  field", nm, sdf.shape)
        except Exception as e: print("  ", e); break
