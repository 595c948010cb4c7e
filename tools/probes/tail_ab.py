"""A/B of the occupancy-grid back-end's fused optimiser tail (Adam + weight average + commit in ONE launch,
NgpConfig.fuse_optimizer_tail) on one box: python tools/probes/tail_ab.py {on|off} -- bench args
(The nerfacto engine's commit is a node of its own, nvo_opt_commit_table: there is no switch to flip for it.)"""
import runpy
import sys

sys.path.insert(0, ".")
mode = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
if mode == "off":
    import nerf_vo_amd.ngp_engine as n

    init = n.NgpConfig.__init__

    def wrapped(self, *a, **kw):
        kw.setdefault("fuse_optimizer_tail", False)
        init(self, *a, **kw)

    n.NgpConfig.__init__ = wrapped
runpy.run_path("bench.py", run_name="__main__")
