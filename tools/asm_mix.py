"""Instruction mix per kernel of a gfx950 assembly listing (hipcc --cuda-device-only -S):
MFMA count, other vector-ALU instructions, packed-f32 arithmetic, AGPR moves, LDS and memory
operations, scratch.  Usage: python tools/asm_mix.py file.s [substring-of-kernel-name ...]"""
import collections
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2:]
    parts = re.split(r"\n(?=_Z[A-Za-z0-9_]*:)", txt)
    for f in parts:
        name = f.split(":", 1)[0]
        if not name.startswith("_Z"):
            continue
        body = f.split(".Lfunc_end", 1)[0]
        ops = collections.Counter(re.findall(r"^\s+([vsdgb][a-z0-9_]+)\s", body, re.M))
        mf = sum(v for k, v in ops.items() if k.startswith("v_mfma"))
        if mf == 0:
            continue
        dn = demangle(name)
        short = re.sub(r"fpn::\(anonymous namespace\)::", "", dn)[:110]
        if want and not any(w in short for w in want):
            continue
        valu = sum(v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        pk32 = {k: v for k, v in ops.items() if re.match(r"v_pk_(add|mul|fma|mov)_", k) and ("f32" in k or "b32" in k)}
        acc = ops.get("v_accvgpr_read_b32", 0) + ops.get("v_accvgpr_write_b32", 0) + ops.get("v_accvgpr_mov_b32", 0)
        scratch = sum(v for k, v in ops.items() if k.startswith("scratch_"))
        m = re.search(r"; NumVgprs: (\d+)", f)
        a = re.search(r"; NumAgprs: (\d+)", f)
        print(short)
        print("   mfma %d  valu %d (%.2f per mfma)  pk32 %s  agpr-moves %d  v_mov %d  ds %d  global %d  scratch %d  salu %d  vgpr %s agpr %s"
              % (mf, valu, valu / mf, pk32, acc, ops.get("v_mov_b32", 0),
                 sum(v for k, v in ops.items() if k.startswith("ds_")),
                 sum(v for k, v in ops.items() if k.startswith("global_") or k.startswith("buffer_")),
                 scratch, sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith("s_waitcnt") and not k.startswith("s_nop")),
                 m.group(1) if m else "?", a.group(1) if a else "?"))
        top = sorted(((v, k) for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma")), reverse=True)[:10]
        print("   " + "  ".join("%s %d" % (k, v) for v, k in top))


if __name__ == "__main__":
    main()
