"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU, exports
every symbol that include/rans4x16_hip.h declares, fails loudly (never silently falls back)
when no device exists, and its host-side arithmetic (the compress bound) matches the oracle."""
import os
import re

import pytest

import htscodecs_amd
from htscodecs_amd import lib as hlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for hdr in ("rans4x16_hip.h", "rans4x8_hip.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(rans4x16_hip_\w+|rans4x8_hip_\w+|rans_\w+_4x16|rans_compress|rans_uncompress)\s*\(", text)
    return sorted(set(names))


def test_library_built_and_loads():
    L = htscodecs_amd.load()
    assert b"gfx950" in L.rans4x16_hip_version()


def test_every_declared_symbol_is_exported_and_bound():
    L = htscodecs_amd.load()
    declared = _declared_symbols()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/rans4x16_hip.h but not exported"
        assert name in hlib.SIGNATURES, f"{name} has no ctypes signature in htscodecs_amd/lib.py"
    for name in hlib.SIGNATURES:
        assert name in declared, f"{name} bound in lib.py but not declared in a header"


def test_exports_are_exactly_the_headers_symbols():
    """-fvisibility=hidden + exports.map: the dynamic symbol table holds the C ABI and nothing else (it
    used to export launchers and mangled C++)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", hlib.LIB_PATH], check=True, stdout=subprocess.PIPE, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == _declared_symbols()


def test_reference_symbols_present():
    L = htscodecs_amd.load()
    for name in ("rans_compress_bound_4x16", "rans_compress_to_4x16", "rans_compress_4x16",
                 "rans_uncompress_to_4x16", "rans_uncompress_4x16"):
        assert hasattr(L, name)


def test_bound_matches_oracle(oracle):
    for size in (0, 1, 20, 21, 1000, 65536, 1 << 20, 1043156, 50_000_000, 2**31 - 1):
        for order in (0, 1, 64, 65, 128, 129, 192, 193, 8, 9, 0x0309, 0xc9, 16, 32):
            assert htscodecs_amd.rans_compress_bound_4x16(size, order) == oracle.bound(size, order)


def test_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert htscodecs_amd.rans_compress_4x16(b"abcabcabcabc" * 10, 0) is None
    assert htscodecs_amd.rans_uncompress_4x16(b"\x00\x05hello") is None
    assert htscodecs_amd.load().rans4x16_hip_create(0) is None


def test_product_does_not_reference_oracle():
    # the oracle is test infrastructure: nothing under htscodecs_amd/ may mention it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "htscodecs_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "liboracle" not in text and "orc_" not in text and "cpu_libs" not in text, fn


REF = "/root/reference"
FIVE = {"rans_compress_bound_4x16", "rans_compress_to_4x16", "rans_compress_4x16",
        "rans_uncompress_to_4x16", "rans_uncompress_4x16"}


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_drivers_link_unchanged_against_the_library(tmp_path):
    """The drop-in claim, kept honest: the reference's own CLI / benchmark driver and its libFuzzer harness are compiled
    from where they lie, UNCHANGED, and linked against librans4x16_hip.so instead of the reference's objects.
    Call sites: tests/rANS_static4x16pr_test.c:155,170,193,206,236,242,270,292; tests/rANS_static4x16pr_fuzz.c:71.
    The harness #includes the reference's .c file by name (fuzz.c:67); an empty file of that name first on the include
    path makes it take the five symbols from the library like any other caller.  Without a GPU the driver must fail
    cleanly (the library has no CPU path)."""
    import subprocess
    libdir = os.path.join(ROOT, "htscodecs_amd")
    inc = tmp_path / "inc"
    (inc / "htscodecs").mkdir(parents=True)
    (inc / "htscodecs" / "rANS_static4x16pr.c").write_text("/* the library under test provides these symbols */\n")
    (tmp_path / "fuzz_main.c").write_text(
        "#include <stdint.h>\n#include <stddef.h>\nint LLVMFuzzerTestOneInput(uint8_t *, size_t);\n"
        "int main(void) { uint8_t b[4] = {0, 1, 2, 3}; return LLVMFuzzerTestOneInput(b, sizeof b); }\n")
    common = ["gcc", "-O1", "-I", str(inc), "-I", REF, "-I", os.path.join(REF, "htscodecs")]
    link = ["-L", libdir, "-lrans4x16_hip", "-Wl,-rpath," + libdir, "-lm", "-lpthread"]
    exe = {}
    for name, srcs in (("driver", [os.path.join(REF, "tests", "rANS_static4x16pr_test.c")]),
                       ("fuzz", [os.path.join(REF, "tests", "rANS_static4x16pr_fuzz.c"), str(tmp_path / "fuzz_main.c")])):
        objs = []
        for src in srcs:
            obj = str(tmp_path / (name + "_" + os.path.basename(src) + ".o"))
            r = subprocess.run(common + ["-c", src, "-o", obj], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            objs.append(obj)
        und = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True, check=True).stdout
        und = {ln.split()[-1].split("@")[0] for ln in und.splitlines() if ln.strip() and not ln.endswith(":")}
        codec = {s for s in und if "rans" in s.lower() or "hts" in s.lower()}
        assert codec and codec <= FIVE, codec                       # nothing of the codec but the five entry points
        exe[name] = str(tmp_path / name)
        r = subprocess.run(["gcc", "-o", exe[name]] + objs + link, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        # every one of them is bound to OUR library at run time
        dyn = subprocess.run(["nm", "-D", "--undefined-only", exe[name]], capture_output=True, text=True, check=True).stdout
        assert codec <= {ln.split()[-1].split("@")[0] for ln in dyn.splitlines() if ln.strip()}
        ldd = subprocess.run(["ldd", exe[name]], capture_output=True, text=True, check=True).stdout
        assert "librans4x16_hip.so" in ldd and libdir in ldd, ldd
    import torch
    if not torch.cuda.is_available():
        # decode mode checks the NULL it gets (test.c:270-272: exit(1)); the stream-compress loop does not (:292-296)
        stream = b"\x20\x05hello"                                   # X_CAT block of five bytes
        src = tmp_path / "in.r4x16"
        src.write_bytes(len(stream).to_bytes(4, "little") + stream)
        r = subprocess.run([exe["driver"], "-d", str(src), str(tmp_path / "out")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1, (r.returncode, r.stderr)           # fails the way the driver fails on NULL, no crash
        assert "no CPU path" in r.stderr


def test_options_by_name_without_a_gpu():
    """include/rans4x16_hip.h part 2b: options are set by name on a context, or - ctx == NULL - as the process-wide
    defaults; the environment only seeds those defaults, once.  Pure host bookkeeping: checked without a GPU."""
    import ctypes as C
    L = htscodecs_amd.load()
    names = []
    while L.rans4x16_hip_option_name(len(names)):
        names.append(L.rans4x16_hip_option_name(len(names)).decode())
    assert len(names) == len(set(names)) >= 25
    header = open(os.path.join(ROOT, "include", "rans4x16_hip.h")).read()
    for n in names:
        assert re.search(r"\b%s\b" % n, header), f"option {n} is not listed in include/rans4x16_hip.h"
    v = C.c_long(-7)
    for n in names:
        assert L.rans4x16_hip_get_option(None, n.encode(), C.byref(v)) == 0
    assert L.rans4x16_hip_get_option(None, b"dec_direct", C.byref(v)) == 0
    before = v.value
    assert L.rans4x16_hip_set_option(None, b"dec_direct", 5) == 0
    assert L.rans4x16_hip_get_option(None, b"dec_direct", C.byref(v)) == 0 and v.value == 5
    assert L.rans4x16_hip_set_option(None, b"dec_direct", before) == 0
    assert L.rans4x16_hip_set_option(None, b"no_such_option", 1) == -1
    assert L.rans4x16_hip_get_option(None, b"no_such_option", C.byref(v)) == -1
    assert L.rans4x16_hip_set_option(None, None, 1) == -1
    # the library reads the environment in exactly one place
    n_getenv = 0
    for fn in os.listdir(os.path.join(ROOT, "htscodecs_amd", "csrc")):
        if fn.endswith((".hip", ".h")):
            n_getenv += len(re.findall(r"\bgetenv\s*\(", open(os.path.join(ROOT, "htscodecs_amd", "csrc", fn)).read()))
    assert n_getenv == 1, n_getenv


def test_weighted_partition_balances_blocks_of_any_size():
    """include/rans4x16_hip.h part 3: rans4x16_hip_partition cuts a batch into contiguous ranges of near-equal bytes.  On
    the heterogeneous batch of bench.py (sizes log-uniform in 4 KiB .. 1 MiB) the largest share must stay within 5 % of
    the mean for 2, 4 and 8 ranks - what the per-rank hetero figures of `bench.py --gpus N` rest on."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from htscodecs_amd import shard
    for world in (2, 4, 8):
        sizes, _, _ = bench.hetero_plan((2 << 30) * world)
        parts = shard.contiguous_partition(sizes, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(sizes)
        assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        shares = [int(sizes[lo:hi].sum()) for lo, hi in parts]
        assert max(shares) * world / sum(shares) <= 1.05, shares
    # and the degenerate shapes: one giant block among small ones, fewer blocks than ranks
    import numpy as np
    parts = shard.contiguous_partition(np.array([10, 10, 10 ** 6, 10, 10]), 3)
    assert sum(hi - lo for lo, hi in parts) == 5
    parts = shard.contiguous_partition(np.array([5, 5]), 8)
    assert sum(hi - lo for lo, hi in parts) == 2 and all(hi >= lo for lo, hi in parts)


def test_cpulist_parser_for_the_numa_feed():
    """include/rans4x16_hip.h part 3: the multi-device calls pin each device's copier threads to the CPUs of the
    device's NUMA node, read from the kernel's cpulist text.  The parser is pure text work: checked here."""
    import ctypes as C
    L = htscodecs_amd.load()

    def parse(text, nbytes=32):
        buf = (C.c_ubyte * nbytes)()
        n = L.rans4x16_hip_cpulist_parse(text.encode(), buf, nbytes)
        cpus = [c for c in range(8 * nbytes) if buf[c >> 3] & (1 << (c & 7))]
        return n, cpus

    assert parse("0-15,32-47\n") == (32, list(range(16)) + list(range(32, 48)))
    assert parse("3") == (1, [3])
    assert parse("0-3,8") == (5, [0, 1, 2, 3, 8])
    assert parse(" 0-1 , 4-5 ,7\n") == (5, [0, 1, 4, 5, 7])
    assert parse("0-3,2-5") == (6, [0, 1, 2, 3, 4, 5])            # overlapping ranges count once
    assert parse("") == (0, []) and parse("\n") == (0, [])        # a node without CPUs
    assert parse("255") == (1, [255])
    for bad in ("256", "5-3", "1,,2", "a", "1-", "-1", "1 2", "0-1000000000000"):
        assert parse(bad)[0] == -1, bad
    # the machine's own lists parse, whatever they are
    import glob
    for path in glob.glob("/sys/devices/system/node/node*/cpulist"):
        with open(path) as f:
            text = f.read()
        n, cpus = parse(text, 1024)
        assert n == len(cpus) and n >= 0, (path, text)
