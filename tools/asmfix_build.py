#!/usr/bin/env python3
"""Tuning aid (not part of the product): build a variant of the HIP library THROUGH its assembly text, so that the text can be checked
and repaired on the way.  Why: with `-mllvm -disable-machine-cse` the ROCm 7.2 backend emits `s_mov_b64 s[a:b], <64-bit literal>` --
an operand gfx950 does not have (its own assembler refuses the line); in the object the literal is cut to its low 32 bits, so
1024.0 = 0x4090000000000000 becomes 0 and the library exp() returns 0 for every negative argument.  This script splits
such lines into two s_mov_b32 and then assembles, links, bundles and embeds the code objects the way hipcc does.

  python tools/asmfix_build.py name "extra flags"        -> tools/_build/librfsgpu_<name>.so
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_build")
LLVM = "/opt/rocm/lib/llvm/bin"
ARCHS = ["xnack+", "xnack-"]
BAD = re.compile(r"^(\s*)s_mov_b64(?:_e32)?\s+s\[(\d+):(\d+)\],\s*(0x[0-9a-fA-F]+|\d+)\s*(;.*)?$")


def repair(text):
    """-> (text, number of lines split).  Any OTHER instruction with a literal wider than 32 bits is an error."""
    out, n = [], 0
    for line in text.split("\n"):
        m = BAD.match(line)
        if m and int(m.group(4), 0) > 0xFFFFFFFF:
            v = int(m.group(4), 0)
            out.append(f"{m.group(1)}s_mov_b32 s{m.group(2)}, 0x{v & 0xFFFFFFFF:x}")
            out.append(f"{m.group(1)}s_mov_b32 s{m.group(3)}, 0x{v >> 32:x}")
            n += 1
            continue
        w = re.search(r"\b0x[0-9a-fA-F]{9,}\b", line.split(";")[0])
        if w and re.match(r"\s+[sv]_\w*64", line):            # (32-bit operands are printed sign-extended: s_mov_b32 s0, 0xfffffffffee00000)
            v = int(w.group(0), 0)
            if not (v >> 32 == 0xFFFFFFFF and v & 0x80000000):
                raise RuntimeError("a literal wider than 32 bits outside s_mov_b64: " + line.strip())
        out.append(line)
    return "\n".join(out), n


def main():
    import __graft_entry__ as g
    bm = g.load_package().build_mod
    name = sys.argv[1]
    extra = sys.argv[2].split() if len(sys.argv) > 2 else []
    os.makedirs(OUT, exist_ok=True)
    work = os.path.join(OUT, "asmfix_" + name)
    os.makedirs(work, exist_ok=True)
    common = [f for f in bm.FLAGS if not f.startswith("--offload-arch") and not f.startswith("-parallel-jobs") and f not in ("-shared",)]
    src = os.path.join(bm.CSRC, "rfsgpu_engine.hip")
    procs = []
    for a in ARCHS:
        s = os.path.join(work, f"dev_{a}.s")
        if os.environ.get("ASMFIX_REUSE") == "1" and os.path.exists(s):      # (the compiler's text from an earlier run)
            procs.append((a, s, subprocess.Popen(["true"])))
            continue
        procs.append((a, s, subprocess.Popen([bm.hipcc(), f"--offload-arch=gfx950:{a}", "--offload-device-only", "-S"] + [f for f in common if f != "-fPIC"] + extra + [src, "-o", s],
                                             stderr=subprocess.DEVNULL)))
    outs = []
    for a, s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("device compile failed for " + a)
        text, n = repair(open(s).read())
        fixed = s.replace(".s", ".fixed.s")
        open(fixed, "w").write(text)
        print(f"{name} {a}: {n} s_mov_b64 lines with a 64-bit literal split", flush=True)
        o, co = fixed.replace(".s", ".o"), fixed.replace(".s", ".out")
        subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu=gfx950:{a}", "-c", fixed, "-o", o])
        subprocess.check_call([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", o, "-o", co])
        outs.append((a, co))
    fb = os.path.join(work, "dev.hipfb")
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
                           "-targets=host-x86_64-unknown-linux-gnu," + ",".join(f"hipv4-amdgcn-amd-amdhsa--gfx950:{a}" for a, _ in outs),
                           "-input=/dev/null"] + [f"-input={co}" for _, co in outs] + [f"-output={fb}"])
    lib = os.path.join(OUT, f"librfsgpu_{name}.so")
    subprocess.check_call([bm.hipcc(), "--offload-arch=gfx950:xnack-", "--offload-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb] + common + ["-shared"] + extra + [src, "-ldl", "-o", lib],
                          stderr=subprocess.DEVNULL)
    print(lib)


if __name__ == "__main__":
    main()
