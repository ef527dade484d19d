"""tools/isa_lint.py: the detector for the spill-placement defect of this toolchain (profiles/r02_exec_restore_spill.md).
CPU-only: the pattern on synthetic gfx950 assembly, and the stamp build() leaves for the library that ships."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

BAD = """
_Z1kv:
\ts_and_saveexec_b64 s[14:15], vcc
\ts_cbranch_execz .LBB0_5
; %bb.4:
\tv_mov_b32_e32 v78, v1
.LBB0_5:
\ts_mov_b32 s36, 0xf800000
\tscratch_store_dword off, v78, off offset:784 ; 4-byte Folded Spill
\ts_or_b64 exec, exec, s[14:15]
\tscratch_load_dwordx4 v[0:3], off, off offset:972 ; 16-byte Folded Reload
\ts_endpgm
"""
GOOD = BAD.replace("\tscratch_store_dword off, v78, off offset:784 ; 4-byte Folded Spill\n\ts_or_b64 exec, exec, s[14:15]\n",
                   "\ts_or_b64 exec, exec, s[14:15]\n\tscratch_store_dword off, v78, off offset:784 ; 4-byte Folded Spill\n")
# a spill at the END of the divergent region (in front of the join label) is legitimate: it belongs to the region's lanes
TAIL = BAD.replace(".LBB0_5:\n\ts_mov_b32 s36, 0xf800000\n\tscratch_store_dword off, v78, off offset:784 ; 4-byte Folded Spill\n",
                   "\tscratch_store_dword off, v78, off offset:784 ; 4-byte Folded Spill\n.LBB0_5:\n\ts_mov_b32 s36, 0xf800000\n")


def test_lint_flags_spill_code_in_front_of_the_exec_restore(tmp_path):
    for name, text, n in (("bad.s", BAD, 1), ("good.s", GOOD, 0), ("tail.s", TAIL, 0)):
        p = os.path.join(str(tmp_path), name)
        open(p, "w").write(text)
        sites = isa_lint.lint(p)
        assert len(sites) == n, (name, sites)
    func, label, line, text = isa_lint.lint(os.path.join(str(tmp_path), "bad.s"))[0]
    assert func == "_Z1kv" and label == ".LBB0_5" and "offset:784" in text


def test_hoisting_the_exec_restore_repairs_the_block(tmp_path):
    p = os.path.join(str(tmp_path), "bad.s")
    open(p, "w").write(BAD)
    assert isa_lint.hoist_exec_restores(p) == 1 and isa_lint.lint(p) == []
    text = open(p).read()
    assert text.index("s_or_b64 exec, exec, s[14:15]") < text.index("s_mov_b32 s36") < text.index("scratch_store_dword off, v78")
    for clean in (GOOD, TAIL):  # nothing to do: files stay as they are
        open(p, "w").write(clean)
        assert isa_lint.hoist_exec_restores(p) == 0 and open(p).read() == clean


def test_the_library_that_ships_passed_the_lint(built):
    """build() compiles with -save-temps, lints the gfx950 assembly of the three translation units and refuses a library with a
    site; the stamp records the result for the sources and flags the library was built from."""
    import __graft_entry__ as g
    stamp = json.load(open(os.path.join(ROOT, "vision_amd", "lib", "isa_lint.json")))
    assert stamp["sites"] == 0 and len(stamp["files"]) == 3, stamp
    assert stamp["build_id"] == g.device_build_id(), "libvmk.so is older than the device sources: run python __graft_entry__.py"
