import sys; sys.path.insert(0,'/root/repo')
import __graft_entry__ as graft, bench, torch
pkg = graft.load_package()
dev = torch.device("cuda", 0); n = 1024
dtype, iso, thr = bench.WORKLOADS["marschner_lobb"]
vol = bench.generate_block(pkg, torch, "marschner_lobb", n, 0, n, None, dev); torch.cuda.synchronize()
ex = pkg.Extractor(0); desc = pkg.make_desc(dtype, (n, n, n))
prm = pkg.make_params(iso, triangles=True, project=True, threshold=thr, step=0.25, relax=0.95, max_steps=50)
for waves in (16384, 8192, 4096):
    ex.debug_option("defaults", 0); ex.debug_option("proj_waves", waves)
    out = {}
    for mode in (2, 3, 4, 5):
        ex.debug_option("proj_literal", mode)
        r = ex.extract_device(vol.data_ptr(), desc, prm)
        out[mode] = int(r.proj_iterations) >> 32
    print("waves", waves, "tail wave-passes", out[2], "all wave-passes", out[3], "tail lane-passes", out[4], "all lane-passes", out[5],
          "tail share %.3f" % (out[2]/out[3]), "mean lanes all %.1f tail %.1f main %.1f" % (out[5]/out[3], out[4]/max(out[2],1), (out[5]-out[4])/(out[3]-out[2])))
