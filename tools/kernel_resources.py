#!/usr/bin/env python3
"""Registers, spills and LDS of every kernel in a built libcrtfx.so, read from the code-object metadata (no GPU needed).

    python tools/kernel_resources.py [path/to/libcrtfx.so] [substring ...]

The library's `.hip_fatbin` section is a run of clang offload bundles, one per translation unit; each holds one gfx950 code object (an ELF
whose NT_AMDGPU_METADATA note lists, per kernel, `.vgpr_count`, `.agpr_count`, `.sgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`,
`.group_segment_fixed_size` (static LDS), `.private_segment_fixed_size` (scratch) and `.max_flat_workgroup_size`).  `resources()` returns
{demangled kernel name: {...}}; tests/test_evidence_tools.py pins the figures the four-blocks-per-CU design of the headline kernels hangs on."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size",
          ".private_segment_fixed_size", ".max_flat_workgroup_size", ".wavefront_size")


def _tool(name):
    p = os.path.join(LLVM_BIN, name)
    return p if os.access(p, os.X_OK) else name


def code_objects(lib_path, arch="gfx950"):
    """The device ELFs of `arch` in the library, as bytes, one per translation unit."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([_tool("llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(td, "unused.o")], check=True)
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if triple.startswith("hip") and arch in triple and size:
                out.append(data[base + off:base + off + size])
    return out


def _demangle(names):
    if not names:
        return {}
    from shutil import which
    filt = next((t for t in (os.path.join(LLVM_BIN, "llvm-cxxfilt"), which("c++filt"), which("llvm-cxxfilt")) if t and os.access(t, os.X_OK)), None)
    if filt is None:
        return {}                                   # no demangler: mangled names are returned
    r = subprocess.run([filt], input="\n".join(names) + "\n", capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.splitlines()))


def resources(lib_path, arch="gfx950"):
    """{demangled kernel name (no 'void ', no argument list): {field without the leading dot: int}}"""
    import yaml
    raw = {}
    for elf in code_objects(lib_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([_tool("llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        m = re.search(r"^\s*---\n(.*?)^\s*\.\.\.\s*$", txt, re.S | re.M)
        if not m:
            continue
        meta = yaml.safe_load(m.group(1))
        for k in meta.get("amdhsa.kernels", []):
            raw[k[".name"]] = {f[1:]: int(k.get(f, 0)) for f in FIELDS}
    dem = _demangle(list(raw))
    out = {}
    for mangled, vals in raw.items():
        name = re.sub(r"^void ", "", dem.get(mangled, mangled))
        depth, cut = 0, len(name)
        for i, ch in enumerate(name):                 # strip the argument list: the last top-level '('
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        out[name[:cut]] = vals
    return out


def main(argv):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = argv[1] if len(argv) > 1 and os.path.exists(argv[1]) else os.path.join(here, "pythoncrt_amd", "libcrtfx.so")
    subs = [a for a in argv[1:] if a != lib]
    res = resources(lib)
    print(f"{'kernel':100s} vgpr agpr sgpr vspill sspill   lds scratch")
    for name in sorted(res):
        if subs and not any(s in name for s in subs):
            continue
        v = res[name]
        print(f"{name[:100]:100s} {v['vgpr_count']:4d} {v['agpr_count']:4d} {v['sgpr_count']:4d} {v['vgpr_spill_count']:6d} {v['sgpr_spill_count']:6d} "
              f"{v['group_segment_fixed_size']:5d} {v['private_segment_fixed_size']:7d}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
