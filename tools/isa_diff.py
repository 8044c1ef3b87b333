#!/usr/bin/env python3
"""Which kernels' gfx950 ISA changed since a commit?      python tools/isa_diff.py <commit> [file.hip ...]

Compiles every mq_det_amd/csrc/*.hip of the working tree and of <commit> (git worktree in a temp dir) to device assembly with the
flags of mq_det_amd/build.py and compares the text per kernel function (debug / path lines and the path-derived __hip_cuid_ symbol
ignored).  Use: after a refactoring that is meant to leave the shipped kernels alone (round 2: the bf16 compile switch, namespaces,
entry-point macros), show that the objects validated on the MI355X in an earlier GPU call are still what is built."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernels(csrc, name, out_dir):
    sys.path.insert(0, ROOT)
    from mq_det_amd import build
    asm = os.path.join(out_dir, name.replace(".hip", ".s"))
    r = subprocess.run([HIPCC, *[f for f in build.FLAGS if f != "-fPIC"], *build.EXTRA_FLAGS.get(name, []), "-S", "--cuda-device-only", name, "-o", asm],
                       cwd=csrc, capture_output=True, text=True)
    if r.returncode:
        return None
    txt = open(asm).read()
    txt = "\n".join(l for l in txt.split("\n") if not re.match(r"\s*(\.file|\.ident|;|\.loc\b)", l) and "__hip_cuid_" not in l)
    funcs = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*\.Lfunc_end\d+:", txt, re.S | re.M):
        body = re.sub(r"\s*;[^\n]*", "", m.group(2))                 # trailing comments
        funcs[m.group(1)] = re.sub(r"\.L\w+", ".L", body)
    return funcs


def main():
    commit = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        wt = os.path.join(tmp, "wt")
        subprocess.check_call(["git", "-C", ROOT, "worktree", "add", "-q", "--detach", wt, commit])
        try:
            new_dir, old_dir = os.path.join(tmp, "new"), os.path.join(tmp, "old")
            os.makedirs(new_dir), os.makedirs(old_dir)
            names = sys.argv[2:] or sorted(f for f in os.listdir(os.path.join(ROOT, "mq_det_amd", "csrc")) if f.endswith(".hip") and f != "api.hip")
            for n in names:
                new = kernels(os.path.join(ROOT, "mq_det_amd", "csrc"), n, new_dir)
                old = kernels(os.path.join(wt, "mq_det_amd", "csrc"), n, old_dir) if os.path.exists(os.path.join(wt, "mq_det_amd", "csrc", n)) else None
                if old is None:
                    print(f"{n:20s} new file ({len(new or {})} kernels)")
                    continue
                changed = sorted(k for k in new if k in old and new[k] != old[k])
                added, removed = sorted(set(new) - set(old)), sorted(set(old) - set(new))
                status = "identical" if not (changed or added or removed) else "CHANGED"
                print(f"{n:20s} {status}: {len(new)} kernels" + (f", changed {changed}" if changed else "") + (f", added {added}" if added else "") +
                      (f", removed {removed}" if removed else ""))
        finally:
            subprocess.call(["git", "-C", ROOT, "worktree", "remove", "--force", wt])


if __name__ == "__main__":
    main()
