"""Instruction mix per kernel of gfx950 code: MFMA count, other vector-ALU instructions,
packed-f32 arithmetic, AGPR moves, LDS / global / flat memory operations, scratch.  Input: an
assembly listing (hipcc --cuda-device-only -S) or a built object / library (its gfx950 code
object is unbundled and disassembled).
Usage: python tools/asm_mix.py file.{s,o,so} [substring-of-kernel-name ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(path):
    """Text of the gfx950 code object inside a host object or shared library, in the
    listing's shape (label lines 'name:', instructions indented)."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "code.co")
        subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], check=True)
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True)
        txt = subprocess.run([LLVM + "llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    txt = re.sub(r"^[0-9a-f]+ <([^>]+)>:", r"\1:", txt, flags=re.M)
    return re.sub(r"\s+// [0-9A-F]+:.*$", "", txt, flags=re.M)


def demangle(n):
    for tool in (LLVM + "llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
        except OSError:
            continue
        if out:
            return out
    return n


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help") or not os.path.isfile(sys.argv[1]):
        sys.exit(__doc__)
    txt = open(sys.argv[1]).read() if sys.argv[1].endswith(".s") else disassemble(sys.argv[1])
    want = sys.argv[2:]
    parts = re.split(r"\n(?=_Z[A-Za-z0-9_]*:)", txt)
    for f in parts:
        name = f.split(":", 1)[0]
        if not name.startswith("_Z"):
            continue
        body = f.split(".Lfunc_end", 1)[0]
        ops = collections.Counter(re.findall(r"^\s+([vsdgbf][a-z0-9_]+)\s", body, re.M))
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
        print("   mfma %d  valu %d (%.2f per mfma)  pk32 %s  agpr-moves %d  v_mov %d  ds %d  global %d  flat %d  scratch %d  salu %d  vgpr %s agpr %s"
              % (mf, valu, valu / mf, pk32, acc, ops.get("v_mov_b32", 0),
                 sum(v for k, v in ops.items() if k.startswith("ds_")),
                 sum(v for k, v in ops.items() if k.startswith("global_") or k.startswith("buffer_")),
                 sum(v for k, v in ops.items() if k.startswith("flat_")), scratch, sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith("s_waitcnt") and not k.startswith("s_nop")),
                 m.group(1) if m else "?", a.group(1) if a else "?"))
        top = sorted(((v, k) for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma")), reverse=True)[:10]
        print("   " + "  ".join("%s %d" % (k, v) for v, k in top))


if __name__ == "__main__":
    main()
