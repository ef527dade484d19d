#!/usr/bin/env python3
"""IR lint for the SECOND miscompile of this toolchain (hipcc 7.2 / AMD clang 22): StructurizeCFG's "hoist zero-cost else-block phi
values" rewrite (profiles/r03_slp_root_cause.md).

What goes wrong.  When a block E ("else") ends in an unconditional branch to S and S has a phi whose incoming value from E is a
zero-cost instruction I of E (an insertelement the SLP vectoriser made, a bitcast, an fneg ...) with no operand defined in E, the
pass moves I up into the block that dominates both sides of the branch, and after it has rebuilt the phis it replaces the phi
incoming `I` by a Flow-block phi `Q = phi [X, then-side], [I, dominator]`.  That is only right when no lane of the then-side goes on
INTO E.  When E has a second predecessor on the then-side (then -> E, the shape of "if (rr) { if (!survive) break; T /= q; }
continue-block"), the lanes that arrive that way now read X — in k_render the ray direction of the PREVIOUS bounce — instead of I.

What this lint does.  It reads the IR of every function as it stands right before StructurizeCFG (`-mllvm
-print-after=unify-loop-exits`, one dump per function) and lists every (E, S, I) that satisfies the pass's own hoisting condition,
from a deliberately WIDER instruction list than the cost model's.  A site is reported when E is the else side of a
conditional branch whose then side can reach E again by another edge (the shape above); every other candidate is the harmless
if/else shape (counted, not reported).  Zero reported sites means
the rewrite had nothing to misapply in that translation unit.

  lint_ir(path) -> (n_functions, harmless, [(function, E, S, instruction text, preds of E)])
  dump_ir(hipcc, flags, tu, out, cwd)   run the device codegen stage of `tu` once more with the dump switched on (cwd holds the
                                        -save-temps=obj files of the build of tu.o, flags include -save-temps=obj)
usage: structurize_lint.py file.ll [...]
"""
import re, shlex, subprocess, sys

# What `opt -passes="print<cost-model>" -cost-kind=latency -mcpu=gfx950` prices at ZERO (the pass hoists only those; table taken with
# tools/experiments/cost_model_probe.ll): insertelement / extractelement of 32-bit lanes at a constant index, extractvalue, every
# shufflevector, fneg, fabs, freeze, zext, trunc to i32, constant-offset getelementptr, ptrtoint / inttoptr / addrspacecast and
# bitcasts that involve a vector (except <2 x float> -> i64).  Priced at 1, hence never hoisted: scalar float <-> int bitcasts,
# insertvalue, sext, fpext / fptrunc, variable-index element access, canonicalize, readfirstlane.  The list below is that table made
# a little wider (any index, any lane width, any getelementptr): one too many costs a false alarm, one too few would hide a site.
FREE_OPS = {"addrspacecast", "inttoptr", "ptrtoint", "getelementptr", "trunc", "zext", "insertelement", "extractelement", "extractvalue",
            "shufflevector", "freeze", "fneg"}
FREE_CALLS = ("@llvm.fabs", "@llvm.ssa.copy", "@llvm.expect", "@llvm.launder", "@llvm.strip", "@llvm.annotation", "@llvm.ptr.annotation")


def _is_free(op, rest):
    if op == "bitcast":  # `bitcast <src type> %v to <dst type>`: free when a vector is involved
        return "<" in rest
    return op in FREE_OPS or (op == "call" and any(c in rest for c in FREE_CALLS))


NAME = r'%(?:"[^"]*"|[-\w.$]+)'
RE_DEF = re.compile(r"^\s+(" + NAME + r") = (?:(?:tail|musttail|notail) )?(\w+)\b(.*)$")
RE_LABEL = re.compile(r'^("[^"]*"|[-\w.$]+):')
RE_INCOMING = re.compile(r"\[ (.*?), (" + NAME + r") \]")
RE_NAME = re.compile(NAME)


def _functions(path):
    """yield (name, [lines]) for every function body in an IR dump (a dump may hold the same function only once)"""
    name, body = None, []
    with open(path, errors="replace") as fh:
        for line in fh:
            if name is None:
                if line.startswith("define "):
                    m = re.search(r"@(\"[^\"]*\"|[-\w.$]+)\(", line)
                    name, body = (m.group(1) if m else "?"), []
            elif line.startswith("}"):
                yield name, body
                name = None
            else:
                body.append(line.rstrip("\n"))


def _lint_function(fname, body):
    blocks, order, cur = {}, [], None  # label -> [instruction lines]
    for line in body:
        m = RE_LABEL.match(line)
        if m:
            cur = "%" + m.group(1); blocks[cur] = []; order.append(cur)
            continue
        if not line.startswith("  ") or line.lstrip().startswith(";"):
            continue
        if cur is None:  # the unlabelled entry block
            cur = "%<entry>"; blocks[cur] = []; order.append(cur)
        blocks[cur].append(line)
    succs, preds, defs = {}, {b: set() for b in blocks}, {}
    for b, ins in blocks.items():
        for line in ins:
            m = RE_DEF.match(line)
            if m:
                defs[m.group(1)] = (b, m.group(2), m.group(3))
        term = ins[-1].strip() if ins else ""
        tg = []
        if term.startswith("br ") or term.startswith("switch ") or term.startswith("indirectbr ") or term.startswith("callbr ") or term.startswith("invoke "):
            tg = re.findall(r"label (" + NAME + r")", term)
        succs[b] = tg
        for t in tg:
            preds.setdefault(t, set()).add(b)
    harmless, sites = 0, []
    for s, ins in blocks.items():
        for line in ins:
            m = RE_DEF.match(line)
            if not m or m.group(2) != "phi":
                break  # phis lead the block
            for val, e in RE_INCOMING.findall(m.group(3)):
                val = val.strip()
                if not val.startswith("%") or val not in defs or e not in blocks:
                    continue
                db, op, rest = defs[val]
                if db != e or succs.get(e) != [s] or not blocks[e][-1].strip().startswith("br label"):
                    continue
                if not _is_free(op, rest):
                    continue
                if any(defs.get(o, (None,))[0] == e for o in RE_NAME.findall(rest) if o != val):
                    continue  # an operand lives in E: the pass leaves it alone
                # the pass treats E as the ELSE side of a predecessor P that branches conditionally to E and to a THEN side `other`
                # (gatherPredicates); the rewrite is unsound when the THEN side can reach E by another edge, without passing P
                shape = None
                for p in preds.get(e, ()):
                    if len(succs[p]) != 2 or e not in succs[p] or succs[p][0] == succs[p][1]:
                        continue
                    other = succs[p][1] if succs[p][0] == e else succs[p][0]
                    seen, todo = {p, e}, [other]
                    while todo and shape is None:
                        b = todo.pop()
                        if b in seen:
                            continue
                        seen.add(b)
                        for t in succs.get(b, ()):
                            if t == e:
                                shape = (p, other, b)
                                break
                            todo.append(t)
                    if shape:
                        break
                if shape:
                    sites.append((fname, e, s, (val + " = " + op + rest).strip()[:160], "branch %s: else %s, then %s reaches it again through %s" % (shape[0], e, shape[1], shape[2])))
                else:
                    harmless += 1
    return harmless, sites


# Sites inside library code this repository does not own (rocPRIM's radix sort, instantiated by the BVH build), reviewed one by one in
# the IR AFTER the pass (`-mllvm -print-after=si-annotate-control-flow -mllvm -filter-print-funcs=<kernel>`): both extractelements
# are still in their block and the loop-header phi still takes them from it, i.e. the pass did not touch them (the successor is a
# loop header outside the structurised region).  (function-name prefix, instruction text without the value name)
REVIEWED_LIBRARY_SITES = (
    ("_ZN7rocprim17ROCPRIM_400200_NS6detail17trampoline_kernelINS1_36wrapped_radix_sort_block_sort_config", "extractelement <2 x i32> %6, i32 0"),
    ("_ZN7rocprim17ROCPRIM_400200_NS6detail17trampoline_kernelINS1_36wrapped_radix_sort_block_sort_config", "extractelement <2 x i32> %6, i32 1"),
)


def is_reviewed_library_site(site):
    fname, _, _, text, _ = site
    return any(fname.startswith(f) and text.split(" = ", 1)[-1].strip() == t for f, t in REVIEWED_LIBRARY_SITES)


def lint_ir(path):
    n, harmless, sites, seen = 0, 0, [], set()
    for fname, body in _functions(path):
        if fname in seen:
            continue
        seen.add(fname); n += 1
        h, s = _lint_function(fname, body)
        harmless += h; sites += s
    return n, harmless, sites


def dump_ir(hipcc, flags, tu, out, cwd):
    """Re-run the gfx950 `-cc1 -S` stage of `hipcc flags tu` (taken from `hipcc -###`, so it is the product's own command line) with
    `-mllvm -print-after=unify-loop-exits`: the IR of every function as StructurizeCFG receives it, written to `out`."""
    import os
    obj = os.path.join(cwd, os.path.splitext(os.path.basename(tu))[0] + ".o")  # -save-temps=obj names the intermediate files after it
    r = subprocess.run([hipcc] + flags + ["-###", "-c", tu, "-o", obj], capture_output=True, text=True, cwd=cwd)
    stages = [shlex.split(l) for l in r.stderr.splitlines() if '"-cc1"' in l and "gfx950" in l and '"-S"' in l or ('"-cc1"' in l and "gfx950" in l and '"-emit-obj"' in l)]
    if not stages:
        raise RuntimeError("structurize_lint: no gfx950 codegen stage in `hipcc -###`:\n" + r.stderr[-2000:])
    a = stages[0]
    j = a.index("-o"); a[j + 1] = "/dev/null"
    a += ["-mllvm", "-print-after=unify-loop-exits"]
    with open(out, "w") as fh:
        subprocess.run(a, stderr=fh, stdout=subprocess.DEVNULL, check=True, cwd=cwd)


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        n, harmless, sites = lint_ir(p)
        print(f"{p}: {n} functions, {harmless} harmless hoist candidates (if/else shape), {len(sites)} site(s) of the unsound shape")
        for s in sites:
            print("  ", s)
        bad += len(sites)
    sys.exit(1 if bad else 0)
