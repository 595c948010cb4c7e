"""A/B of the optimiser tails (commit inside the Adam launch; NGP: + weight average) on one box:
python tools/probes/tail_ab.py {on|off} -- bench args"""
import runpy
import sys

sys.path.insert(0, ".")
mode = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
if mode == "off":
    import nerf_vo_amd.engine as e
    import nerf_vo_amd.ngp_engine as n

    def patch(cls, **defaults):
        init = cls.__init__

        def wrapped(self, *a, **kw):
            for k, v in defaults.items():
                kw.setdefault(k, v)
            init(self, *a, **kw)
        cls.__init__ = wrapped

    patch(e.EngineConfig, commit_in_adam=False)
    patch(n.NgpConfig, fuse_optimizer_tail=False)
runpy.run_path("bench.py", run_name="__main__")
