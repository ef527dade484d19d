#!/usr/bin/env python3
"""ISA lint for the spill-placement defect of this toolchain (profiles/r02_exec_restore_spill.md).

hipcc 7.2 / gfx950 can place a register spill or reload — or any other vector instruction — at the top of a control-flow JOIN block
*in front of* the instruction that restores EXEC for the lanes that skipped the divergent region (`s_or_b64 exec, exec, s[a:b]`).
The instruction then runs for the lanes of the region only; the other lanes' slot (or register) keeps a stale value, and whatever
reads it after the join computes with last iteration's data.  Both miscompiles of round 2 are this pattern.

Three tools:
  lint(file.s)                  every VECTOR instruction (VALU, VMEM, scratch, LDS; `kind` says whether it is marked as a spill) that sits
                                between a label that is the target of an `s_cbranch_execz` skip edge (a join block) and the first
                                `s_or_b64 exec, exec, ...` of that block.  Scalar instructions and v_readlane / v_writelane (SGPR spills
                                through lanes: they do not look at EXEC) are not reported.
  hoist_exec_restores(file.s)   the repair: move the EXEC restore to the top of every flagged block — after CHECKING that nothing it
                                jumps over defines the restore's mask registers, touches EXEC, or consumes the SCC the restore clobbers.
                                A block that fails the check raises UnsafeHoist and the build stops.
  lint_binary(libvmk.so)        the same scan on the code objects of the library that SHIPS (extracted with llvm-objdump --offloading,
                                disassembled with --symbolize-operands), so that the verdict is about the installed binary, not about
                                an assembly file that may or may not have gone into it.
usage: isa_lint.py file.s [...] | isa_lint.py --binary libvmk.so      exit status 1 if anything is found."""
import os
import re
import subprocess
import sys
import tempfile

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")
BRANCH_T = re.compile(r"\bs_cbranch_execz\s+(\.LBB\d+_\d+)")
ANY_BRANCH = re.compile(r"^\s*(s_cbranch|s_branch|s_setpc|s_endpgm|s_swappc)")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*s\[(\d+):(\d+)\]")
SPILL = re.compile(r"^\s*scratch_(store|load)_\w+.*(Folded Spill|Folded Reload)|^\s*(v_accvgpr_write|v_accvgpr_read).*(Spill|Reload)")
# vector instructions: they execute under EXEC.  (v_readlane / v_writelane ignore EXEC; v_readfirstlane reads a uniform value.)
VECTOR = re.compile(r"^\s*(v_(?!readlane|writelane|readfirstlane)\w+|scratch_\w+|global_\w+|flat_\w+|buffer_\w+|ds_\w+)\b")
INSTR = re.compile(r"^\s*([a-z_][\w.]*)\s*(.*?)\s*(?:;.*|//.*)?$")
EXEC_WRITE = re.compile(r"^\s*(s_\w*saveexec\w*\s|s_\w+\s+exec(_lo|_hi)?\s*,)")  # any scalar instruction whose destination is EXEC
SCC_READERS = re.compile(r"^\s*(s_cselect_|s_addc_|s_subb_|s_cbranch_scc|s_cmov_)")


class UnsafeHoist(RuntimeError):
    pass


def _sgprs(operand):
    """SGPR numbers named by an operand: s5, s[4:5], vcc (106/107), ..."""
    operand = operand.strip()
    m = re.fullmatch(r"s(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _dest_operands(line):
    """Destination operands of one instruction, as written (conservative: the first operand of anything that has one; VOP3 compares
    and v_div_scale / v_add_co write a second, scalar destination)."""
    m = INSTR.match(line)
    if not m:
        return []
    op, rest = m.group(1), m.group(2)
    if not rest or op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_endpgm", "s_sleep", "s_setprio", "s_cmp", "s_bitcmp")):
        return []
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest)]
    dests = ops[:1]
    if op.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64", "v_mad_i64")) and len(ops) > 1:
        dests.append(ops[1])
    return dests


def _block_span(lines, i):
    """For a join label at line i: (j, restore_match) where j is the line of the block's EXEC restore, or (None, None) when the block
    has none before its first branch / the next label."""
    j = i + 1
    while j < len(lines):
        t = lines[j]
        if LABEL.match(t) or ANY_BRANCH.match(t):
            return None, None
        m = EXEC_RESTORE.match(t)
        if m:
            return j, m
        if EXEC_WRITE.match(t):
            return None, None  # the block opens another divergent region (s_or_saveexec / s_and_saveexec / s_xor exec ...): not a join prologue
        j += 1
    return None, None


def lint(path):
    lines = open(path, errors="replace").read().split("\n")
    join_targets = {m.group(1) for l in lines for m in [BRANCH_T.search(l)] if m}
    found, func = [], "?"
    for i, l in enumerate(lines):
        fm = FUNC.match(l)
        if fm and not l.startswith(".L"):
            func = fm.group(1)
        m = LABEL.match(l)
        if not (m and m.group(1) in join_targets):
            continue
        j, _ = _block_span(lines, i)
        if j is None:
            continue
        for k in range(i + 1, j):
            t = lines[k]
            if SPILL.match(t):
                found.append((func, m.group(1), k + 1, t.strip(), "spill"))
            elif VECTOR.match(t):
                found.append((func, m.group(1), k + 1, t.strip(), "vector"))
    return found


def check_hoist(span, restore_match, where=""):
    """The EXEC restore `s_or_b64 exec, exec, s[a:b]` may be moved in front of `span` (the instructions between the join label and the
    restore) only if none of them (1) writes s[a] .. s[b] — e.g. a v_readlane_b32 that reloads the saved mask itself —, (2) reads or
    writes EXEC, (3) consumes SCC, which the hoisted s_or_b64 would have clobbered first."""
    mask = set(range(int(restore_match.group(1)), int(restore_match.group(2)) + 1))
    for t in span:
        if not INSTR.match(t) or not t.strip() or t.strip().startswith((";", ".", "//")):
            continue
        body = re.sub(r"(;|//).*$", "", t)
        for d in _dest_operands(t):
            if _sgprs(d) & mask:
                raise UnsafeHoist(f"{where}: `{t.strip()}` defines the mask s[{min(mask)}:{max(mask)}] of the EXEC restore it sits in front of")
        if re.search(r"\bexec(_lo|_hi)?\b", body):
            raise UnsafeHoist(f"{where}: `{t.strip()}` touches EXEC in front of the EXEC restore")
        if SCC_READERS.match(t):
            raise UnsafeHoist(f"{where}: `{t.strip()}` consumes SCC in front of the EXEC restore (s_or_b64 clobbers it)")


def hoist_exec_restores(path):
    """Repair: in every join block the lint flags, move the `s_or_b64 exec, exec, ...` to the top of the block.  Whatever sits between a
    join label and its EXEC restore was put there by passes that assume the join's lanes (the restore itself is emitted first in the
    block by SILowerControlFlow; its operand is computed before the branch), so running the restore first is what the compiler meant —
    provided the restore does not depend on what it jumps over: check_hoist refuses the cases where it does.
    Returns the number of blocks changed; the file is rewritten in place."""
    lines = open(path, errors="replace").read().split("\n")
    join_targets = {m.group(1) for l in lines for m in [BRANCH_T.search(l)] if m}
    changed, func = 0, "?"
    i = 0
    while i < len(lines):
        fm = FUNC.match(lines[i])
        if fm and not lines[i].startswith(".L"):
            func = fm.group(1)
        m = LABEL.match(lines[i])
        if m and m.group(1) in join_targets:
            j, rm = _block_span(lines, i)
            if j is not None and any(SPILL.match(t) or VECTOR.match(t) for t in lines[i + 1:j]):
                check_hoist(lines[i + 1:j], rm, f"{os.path.basename(path)} {func[:60]} {m.group(1)}")
                lines.insert(i + 1, lines.pop(j))
                changed += 1
        i += 1
    if changed:
        open(path, "w").write("\n".join(lines))
    return changed


# ---- the binary that ships ----
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
D_LABEL = re.compile(r"^[0-9a-f]+ <(L\d+)>:")
D_FUNC = re.compile(r"^[0-9a-f]+ <([^L][^>]*|L[^\d>][^>]*)>:")
D_BRANCH_T = re.compile(r"\bs_cbranch_execz\s+(L\d+)\b")
D_ANY_BRANCH = re.compile(r"^\s*(s_cbranch|s_branch|s_setpc|s_endpgm|s_swappc)")


def lint_disassembly(text):
    """The scan of lint() on `llvm-objdump -d --symbolize-operands` output (labels are numbered per function)."""
    found = []
    funcs, cur = [], None
    for l in text.split("\n"):
        fm = D_FUNC.match(l)
        if fm:
            cur = (fm.group(1), [])
            funcs.append(cur)
        elif cur is not None:
            cur[1].append(l)
    for name, lines in funcs:
        targets = {m.group(1) for l in lines for m in [D_BRANCH_T.search(l)] if m}
        for i, l in enumerate(lines):
            m = D_LABEL.match(l)
            if not (m and m.group(1) in targets):
                continue
            pending = []
            for t in lines[i + 1:]:
                if D_LABEL.match(t) or D_ANY_BRANCH.match(t):
                    pending = []
                    break
                if EXEC_RESTORE.match(t):
                    break
                if EXEC_WRITE.match(t):
                    pending = []
                    break
                if VECTOR.match(t):
                    pending.append(t.strip())
            else:
                pending = []
            found += [(name, m.group(1), t) for t in pending]
    return found


def lint_binary(so_path):
    """Extract every gfx950 code object of the shared library, disassemble it and scan it.  Returns (number of code objects, sites)."""
    tmp = tempfile.mkdtemp(prefix="isa_lint_bin_")
    try:
        local = os.path.join(tmp, os.path.basename(so_path))
        import shutil
        shutil.copy(so_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, capture_output=True)
        objs = sorted(f for f in os.listdir(tmp) if "gfx950" in f)
        sites = []
        for f in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            sites += [(f,) + s for s in lint_disassembly(dis)]
        return len(objs), sites
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    bad = 0
    if len(sys.argv) > 2 and sys.argv[1] == "--binary":
        n, sites = lint_binary(sys.argv[2])
        for s in sites:
            print(*s)
        print(f"{sys.argv[2]}: {n} code object(s), {len(sites)} site(s)")
        sys.exit(1 if sites or not n else 0)
    for p in sys.argv[1:]:
        f = lint(p)
        for func, label, ln, text, kind in f:
            print(f"{p}:{ln}: {func[:60]} {label}: {kind} instruction in front of the EXEC restore of a join block: {text}")
        print(f"{p}: {len(f)} site(s)")
        bad += len(f)
    sys.exit(1 if bad else 0)
