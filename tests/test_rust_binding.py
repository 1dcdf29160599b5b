"""The Rust FFI crate under rust/ cannot be compiled here (no Rust toolchain); this test keeps its layout claims honest:
every `assert_size!` / `assert_offset!` line of rust/pt_amd_sys/src/lib.rs must state what the C header gives when
compiled, every function the header declares for the hot path must be declared with the same arity, and every
constant must have the header's value."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
RUST = (ROOT / "rust" / "pt_amd_sys" / "src" / "lib.rs").read_text()
HEADER = (ROOT / "include" / "pt_amd.h").read_text()
HEADER_CODE = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)


def c_layout(queries):
    """Compile a probe against include/pt_amd.h and return the numbers it prints."""
    import tempfile
    lines = "\n".join(f'    printf("%zu\\n", (size_t)({q}));' for q in queries)
    src = f'#include <stdio.h>\n#include <stddef.h>\n#include "pt_amd.h"\nint main(void) {{\n{lines}\n    return 0;\n}}\n'
    with tempfile.TemporaryDirectory() as d:
        c = Path(d) / "probe.c"
        c.write_text(src)
        exe = Path(d) / "probe"
        subprocess.run(["gcc", "-std=c11", f"-I{ROOT / 'include'}", str(c), "-o", str(exe)], check=True)
        out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return [int(x) for x in out]


def test_sizes_and_offsets_match_the_header():
    sizes = re.findall(r"^assert_size!\((\w+), (\d+)\);", RUST, re.M)
    offsets = re.findall(r"^assert_offset!\((\w+), (\w+), (\d+)\);", RUST, re.M)
    assert len(sizes) >= 10 and len(offsets) >= 30
    got = c_layout([f"sizeof({t})" for t, _ in sizes] + [f"offsetof({t}, {f})" for t, f, _ in offsets])
    want = [int(n) for _, n in sizes] + [int(n) for _, _, n in offsets]
    assert got == want


def test_struct_fields_are_in_header_order():
    for name in ("PtObject", "PtMesh", "PtVertex", "PtMaterial", "PtTexture", "PtSettings", "PtFrame", "PtConfig", "PtRay", "PtHit", "PtMultiConfig"):
        c_body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HEADER, re.S).group(1)
        c_body = re.sub(r"/\*.*?\*/", "", c_body, flags=re.S)
        c_fields = re.findall(r"(\w+)(?:\[\d+\])?;", c_body)
        r_body = re.search(r"pub struct %s \{(.*?)\n\}" % name, RUST, re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", r_body)
        assert r_fields == c_fields, name


def test_functions_have_the_header_arity():
    decls = dict(re.findall(r"pub fn (pt_\w+)\((.*?)\) -> ", RUST, re.S))
    for fn in ("pt_create", "pt_destroy", "pt_last_error", "pt_upload_meshes", "pt_upload_materials", "pt_upload_objects",
               "pt_render", "pt_render_device", "pt_render_device_noalloc", "pt_render_submit", "pt_render_submit_device", "pt_render_wait", "pt_render_stream_wait", "pt_render_slot_wait_for", "pt_host_alloc", "pt_host_free", "pt_query", "pt_read_aux", "pt_device_info",
               "pt_multi_create", "pt_multi_destroy", "pt_multi_last_error", "pt_multi_upload_meshes", "pt_multi_upload_materials",
               "pt_multi_upload_objects", "pt_multi_render", "pt_multi_context", "pt_multi_rccl_version"):
        assert fn in decls, fn
        c_args = re.search(r"^(?:int|const char\*) %s\((.*?)\);" % fn, HEADER_CODE, re.S | re.M).group(1)
        assert len(decls[fn].split(",")) == len(c_args.split(",")), fn


def test_constants_match_the_header():
    for name, value in re.findall(r"pub const (PT_\w+): \w+ = (-?[0-9A-Fa-fx_]+);", RUST):
        value = int(value.replace("_", ""), 0)
        m = re.search(r"#define %s\s+(\S+)" % name, HEADER) or re.search(r"\b%s = (-?\w+)" % name, HEADER)
        assert m, name
        assert int(m.group(1).rstrip("u,"), 0) == value, name
