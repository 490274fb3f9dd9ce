"""Packed-fp32 instructions of the built library, by kernel and op_sel form.

Why: on MI355X `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]` (BOTH results read the HIGH half of src1) was seen to return a
low result of 0 in lanes 48-63, now and then, when a second wave shares the SIMD (round 6, DESIGN 7: the mu-zero form of
k_painn_fwd_mma; one block per CU, or the same kernel without packed ops: never).  painn_mma.hip is therefore compiled
without packed fp32 arithmetic; this tool lists what the other kernels contain and tests/test_round6_cpu.py holds the
matrix-pipe PaiNN kernels to zero.

    python tools/scan_packed_opsel.py [libgeossl_hip.so]          # table: kernel, packed ops, by op_sel form
"""
import collections, os, re, struct, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """The gfx950 ELF images inside a HIP fat binary (clang offload bundles: magic, u64 count, {offset, size, triple})."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        b = data.find(MAGIC, pos)
        if b < 0:
            return out
        n = struct.unpack_from("<Q", data, b + len(MAGIC))[0]
        p = b + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                out.append(data[b + off:b + off + size])
        pos = b + len(MAGIC)


def disassemble(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout


PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\b(.*)")


def scan(path):
    """{kernel: Counter(form -> count)}; form = mnemonic + its op_sel / op_sel_hi modifiers ('plain' without op_sel)."""
    table = collections.defaultdict(collections.Counter)
    for image in code_objects(path):
        kernel = None
        for line in disassemble(image).splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                kernel = m.group(1)
                continue
            m = PK.search(line)
            if m and kernel:
                mods = " ".join(re.findall(r"op_sel(?:_hi)?:\[[01,]+\]", m.group(2)))
                table[kernel][m.group(1) + (" " + mods if mods else "")] += 1
    return table


def lo_select_forms(counter):
    """The forms whose LOW result reads a source's high half (an op_sel bit set)."""
    return {f: c for f, c in counter.items() if re.search(r"op_sel:\[[01,]*1", f)}


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "geossl_amd", "lib", "libgeossl_hip.so")
    for k, c in sorted(scan(lib).items()):
        sel = lo_select_forms(c)
        print("%-90s packed %5d  op_sel %4d  %s" % (k[:90], sum(c.values()), sum(sel.values()), dict(sel) if sel else ""))
