#!/usr/bin/env python3
"""ISA lint for the spill-placement defect of this toolchain (profiles/r02_exec_restore_spill.md).

hipcc 7.2 / gfx950 can place a register spill or reload at the top of a control-flow JOIN block *in front of* the instruction that
restores EXEC for the lanes that skipped the divergent region (`s_or_b64 exec, exec, s[..]`).  The store / load then runs for the
lanes of the region only; the other lanes' slot (or register) keeps a stale value, and whatever reads it after the join computes with
last iteration's data.  Both miscompiles of round 2 are this pattern.

The lint walks gfx950 assembly (hipcc -S --cuda-device-only, or llvm-objdump -d of a code object) and reports every
`scratch_{store,load}` that sits between a label that is the target of an `s_cbranch_execz` skip edge (a join
block) and the first `s_or_b64 exec, exec, ...` of that block.   usage: isa_lint.py file.s [...]   exit status 1 if anything is found."""
import re, sys

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")
BRANCH_T = re.compile(r"\bs_cbranch_execz\s+(\.LBB\d+_\d+)")
ANY_BRANCH = re.compile(r"^\s*(s_cbranch|s_branch|s_setpc|s_endpgm|s_swappc)")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
SPILL = re.compile(r"^\s*scratch_(store|load)_\w+.*(Folded Spill|Folded Reload)|^\s*(v_accvgpr_write|v_accvgpr_read).*(Spill|Reload)")


def lint(path):
    lines = open(path, errors="replace").read().split("\n")
    join_targets = set()
    for l in lines:
        m = BRANCH_T.search(l)
        if m:
            join_targets.add(m.group(1))
    found = []
    func = "?"
    i = 0
    n = len(lines)
    while i < n:
        l = lines[i]
        fm = FUNC.match(l)
        if fm and not l.startswith(".L"):
            func = fm.group(1)
        m = LABEL.match(l)
        if m and m.group(1) in join_targets:
            # scan the block prologue: up to the exec restore; stop at any branch / next label
            j = i + 1
            pending = []
            while j < n:
                t = lines[j]
                if LABEL.match(t) or ANY_BRANCH.match(t):
                    pending = []  # no exec restore in this block: not a join of the kind we look for
                    break
                if EXEC_RESTORE.match(t):
                    break
                if SPILL.match(t):
                    pending.append((j + 1, t.strip()))
                j += 1
            else:
                pending = []
            for ln, text in pending:
                found.append((func, m.group(1), ln, text))
        i += 1
    return found


def hoist_exec_restores(path):
    """Repair: in every join block the lint flags, move the `s_or_b64 exec, exec, ...` to the top of the block.  Whatever sits between a
    join label and its EXEC restore was put there by passes that assume the join's lanes (the restore itself is emitted first in the
    block by SILowerControlFlow; its operand is computed before the branch), so running the restore first is what the compiler meant.
    Returns the number of blocks changed; the file is rewritten in place."""
    lines = open(path, errors="replace").read().split("\n")
    join_targets = {m.group(1) for l in lines for m in [BRANCH_T.search(l)] if m}
    changed = 0
    i = 0
    while i < len(lines):
        m = LABEL.match(lines[i])
        if m and m.group(1) in join_targets:
            j = i + 1
            spill = False
            while j < len(lines) and not LABEL.match(lines[j]) and not ANY_BRANCH.match(lines[j]) and not EXEC_RESTORE.match(lines[j]):
                spill = spill or bool(SPILL.match(lines[j]))
                j += 1
            if spill and j < len(lines) and EXEC_RESTORE.match(lines[j]):
                lines.insert(i + 1, lines.pop(j))
                changed += 1
        i += 1
    if changed:
        open(path, "w").write("\n".join(lines))
    return changed


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        f = lint(p)
        for func, label, ln, text in f:
            print(f"{p}:{ln}: {func[:60]} {label}: spill code in front of the EXEC restore of a join block: {text}")
        print(f"{p}: {len(f)} site(s)")
        bad += len(f)
    sys.exit(1 if bad else 0)
