"""TEST INFRASTRUCTURE ONLY.  `python tests/simt/oob_check.py <mode> <check> [<check> ...]`: run parity checks on the emulated kernels
with every library argument against a guard page (tests/simt/guard.py); prints `OK <check>` per check.  A kernel that reads or writes
outside one of its buffers on the guarded side ends this process with SIGSEGV -- the caller (tests/test_simt_kernels_cpu.py) looks at
the return code."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    import torch
    import simt
    from simt import guard
    import parity_checks as pc
    import gdino_checks as gc
    from mq_det_amd.modeling import detector, pipeline
    mode, names = sys.argv[1], sys.argv[2:]
    cpu = torch.device("cpu")

    def prepare(self, device=None):
        self._validate_config()
        self._plan = pipeline.build_plan(self.state_dict(), self.cfg, cpu, dtype=detector.compute_dtype(self.cfg))
        self._plan_key, self.use_hip_graph = cpu, False
        return self._plan
    detector.GeneralizedVLRCNN_New.prepare = prepare
    pc.QUICK, pc.PINS = True, False
    extra = {
        "attention_small": lambda: [pc.check_attention(cpu, B=2, H=3, D=64, Nq=70, Nk=141, mask=True, kvlen=True),
                                    pc.check_attention(cpu, B=1, H=2, D=32, Nq=37, Nk=61),
                                    pc.check_attention(cpu, B=1, H=2, D=32, Nq=37, Nk=700, nsplit=2),
                                    pc.check_attention(cpu, B=1, H=2, D=64, Nq=130, Nk=257, mask=True, clamp=50000.0)],
    }
    with simt.installed(), guard.pointer_guard(mode), guard.guarded_ops(mode):
        for n in names:
            fn = extra.get(n) or (lambda n=n: getattr(pc, n, None)(cpu) if hasattr(pc, n) else getattr(gc, n)(cpu))
            res = fn()
            res = res if isinstance(res, list) else [res]
            bad = [r["name"] for r in res if not r["ok"] and "HIP-graph" not in r["name"]]
            print(("OK " if not bad else "MISMATCH ") + n + (" " + "; ".join(bad[:3]) if bad else ""), flush=True)


if __name__ == "__main__":
    main()
