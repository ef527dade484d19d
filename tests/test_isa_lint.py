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
    func, label, line, text, kind = isa_lint.lint(os.path.join(str(tmp_path), "bad.s"))[0]
    assert func == "_Z1kv" and label == ".LBB0_5" and "offset:784" in text and kind == "spill"


def test_hoisting_the_exec_restore_repairs_the_block(tmp_path):
    p = os.path.join(str(tmp_path), "bad.s")
    open(p, "w").write(BAD)
    assert isa_lint.hoist_exec_restores(p) == 1 and isa_lint.lint(p) == []
    text = open(p).read()
    assert text.index("s_or_b64 exec, exec, s[14:15]") < text.index("s_mov_b32 s36") < text.index("scratch_store_dword off, v78")
    for clean in (GOOD, TAIL):  # nothing to do: files stay as they are
        open(p, "w").write(clean)
        assert isa_lint.hoist_exec_restores(p) == 0 and open(p).read() == clean


# a reload of the saved mask itself in front of its restore — hoisting would restore EXEC from a stale pair
MASK_RELOAD = BAD.replace("\ts_mov_b32 s36, 0xf800000\n", "\tv_readlane_b32 s14, v40, 4\n\tv_readlane_b32 s15, v40, 5\n")
# any vector instruction in front of the restore is reported, not only the ones marked as spills
PLAIN_VALU = BAD.replace("\tscratch_store_dword off, v78, off offset:784 ; 4-byte Folded Spill\n", "\tv_mov_b32_e32 v69, v53\n")
# a block that opens another divergent region before any restore is not a join prologue (the v_cmp belongs to the new region)
NEW_REGION = BAD.replace("\tscratch_store_dword off, v78, off offset:784 ; 4-byte Folded Spill\n", "\ts_or_saveexec_b64 s[0:1], s[12:13]\n\ts_xor_b64 exec, exec, s[0:1]\n\tv_cmp_ne_u32_e32 vcc, 0, v8\n")
SCC_USER = BAD.replace("\ts_mov_b32 s36, 0xf800000\n", "\ts_cselect_b32 s36, s1, s2\n")


def test_hoist_refuses_what_it_cannot_prove_safe(tmp_path):
    import pytest
    p = os.path.join(str(tmp_path), "x.s")
    for text, what in ((MASK_RELOAD, "defines the mask"), (SCC_USER, "consumes SCC")):
        open(p, "w").write(text)
        assert len(isa_lint.lint(p)) == 1
        with pytest.raises(isa_lint.UnsafeHoist, match=what):
            isa_lint.hoist_exec_restores(p)
        assert open(p).read() == text  # untouched
    open(p, "w").write(PLAIN_VALU)
    sites = isa_lint.lint(p)
    assert len(sites) == 1 and sites[0][4] == "vector"
    assert isa_lint.hoist_exec_restores(p) == 1 and isa_lint.lint(p) == []
    open(p, "w").write(NEW_REGION)
    assert isa_lint.lint(p) == [] and isa_lint.hoist_exec_restores(p) == 0


def test_the_scan_of_a_disassembled_code_object():
    dis = """
0000000000001000 <_Z1kv>:
\ts_and_saveexec_b64 s[14:15], vcc                           // 000000001000: BE8E206A
\ts_cbranch_execz L0                                         // 000000001004: BF880002
\tv_mov_b32_e32 v78, v1                                      // 000000001008: 7E9C0301
000000000000100c <L0>:
\ts_mov_b32 s36, 0xf800000                                   // 00000000100C: BEA400FF 0F800000
\tscratch_store_dword off, v78, off offset:784               // 000000001014: DC704310 007C4E00
\ts_or_b64 exec, exec, s[14:15]                              // 00000000101C: 87FE0E7E
\ts_endpgm                                                   // 000000001020: BF810000
"""
    sites = isa_lint.lint_disassembly(dis)
    assert len(sites) == 1 and sites[0][0] == "_Z1kv" and sites[0][1] == "L0" and "scratch_store_dword" in sites[0][2]
    fixed = dis.replace("\tscratch_store_dword off, v78, off offset:784               // 000000001014: DC704310 007C4E00\n", "")
    assert isa_lint.lint_disassembly(fixed) == []


def test_the_library_that_ships_passed_the_lint(built):
    """build() compiles with -save-temps, lints the gfx950 assembly of the three translation units and refuses a library with a
    site; the stamp records the result for the sources and flags the library was built from."""
    import __graft_entry__ as g
    stamp = json.load(open(os.path.join(ROOT, "vision_amd", "lib", "isa_lint.json")))
    assert stamp["sites"] == 0 and len(stamp["files"]) == 3, stamp
    assert stamp["binary"] == {"code_objects": 3, "sites": 0}, stamp  # the scan of the code objects extracted from libvmk.so itself
    assert stamp["build_id"] == g.device_build_id(), "libvmk.so is older than the device sources: run python __graft_entry__.py"
