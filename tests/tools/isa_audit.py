"""Disassemble the gfx950 code objects embedded in libs2p_hip.so (no GPU needed) and tabulate, per kernel, its LDS (DS)
instructions, whether it stages through LDS-DMA, and its packed-fp32 / op_sel instructions (the hazard class of DESIGN.md section 4).  Used by tests/test_host_logic.py::test_lds_access_widths and as a tool:
    python tests/tools/isa_audit.py [path/to/libs2p_hip.so]"""
import collections, os, re, struct, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin_section(so_path):
    """Bytes of the .hip_fatbin section (parsed straight from the ELF64 section table)."""
    data = open(so_path, "rb").read()
    assert data[:4] == b"\x7fELF" and data[4] == 2, "not an ELF64 file"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        secs.append((name, off, size))
    stroff = secs[shstrndx][1]
    for name, off, size in secs:
        end = data.index(b"\0", stroff + name)
        if data[stroff + name:end] == b".hip_fatbin":
            return data[off:off + size]
    raise RuntimeError("no .hip_fatbin section in " + so_path)


def code_objects(so_path, arch="gfx950"):
    """The device ELF of every translation unit bundled into the library."""
    fat = _fatbin_section(so_path)
    out, pos = [], 0
    while True:
        pos = fat.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", fat, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tsz].decode()
            p += 24 + tsz
            if arch in triple and size:
                out.append(fat[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def audit(so_path):
    """{kernel name: {"ds": Counter(DS mnemonic -> count), "lds_dma": int, "lds_bytes": int}}"""
    res = {}
    for co in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True,
                                 check=True).stdout
            meta = subprocess.run([LLVM + "/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        lds = dict(re.findall(r"\.group_segment_fixed_size:\s*(\d+)[^.]*?(?:\.[a-z_]+:[^\n]*\n\s*)*?\.name:\s*(\S+)", meta))
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                if not cur.endswith(".kd"):
                    res.setdefault(cur, {"ds": collections.Counter(), "lds_dma": 0, "packed": collections.Counter()})
                continue
            if cur is None or cur not in res:
                continue
            ins = line.split()
            if not ins:
                continue
            op = ins[0]
            if re.match(r"v_pk_\w+_f32", op) or "op_sel" in line:
                res[cur]["packed"][op + (" op_sel" if "op_sel" in line else "")] += 1
            if op.startswith("ds_"):
                res[cur]["ds"][op] += 1
            elif (op.startswith("buffer_load") or op.startswith("global_load_lds")) and (" lds" in line or op.startswith("global_load_lds")):
                res[cur]["lds_dma"] += 1
    return res


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.splitlines()
        if len(out) == len(names):
            return dict(zip(names, out))
    except OSError:
        pass
    return {n: n for n in names}


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "..", "s2p_amd", "csrc", "libs2p_hip.so")
    a = audit(so)
    dm = demangle(list(a))
    for k in sorted(a, key=lambda k: dm[k]):
        v = a[k]
        if v["ds"] or v["lds_dma"]:
            print("%-90s dma=%-3d %s" % (dm[k][:90], v["lds_dma"], dict(v["ds"])))
