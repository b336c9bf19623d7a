#!/usr/bin/env python3
"""How much of the restatement's text do the reference's fixtures pin?  A mutation score for oracle/npo_*.h (CPU only).

The device code (nuclear_sim_amd/csrc/npd_*.h) and the CPU restatement (oracle/npo_*.h) are, for most subsystems, one text: a
HIP-vs-oracle test checks compiler, staging and scheduling, and what says that the TEXT is the reference's arithmetic is the set of
reference-generated fixtures under tests/golden/.  This tool measures that: it makes single-token mutants of the restatement --

    cmp     a comparison flipped at its boundary or reversed          <  <->  <=     >  <->  >=     ==  <->  !=     <  ->  >
    sign    the sign of a term                                         a + b  <->  a - b
    const   a numeric literal scaled by (1 + 1e-3)                     (clamp bounds, rate constants, thresholds, exponents)
    minmax  a clip's direction                                         npo_pymax <-> npo_pymin
    branch  a branch dropped / forced                                  if (c)  ->  if (0 && (c))   and   if (1 || (c))

-- builds each as its own libnpo.so and replays the fixtures against it (tests/test_oracle_golden.py, tests/test_scenarios.py,
tests/test_statelog_cpu.py: trajectories, initial states, event counts, the reference's own state logs).  A mutant the fixtures
reject is KILLED; one that passes SURVIVES: either an equivalent mutant (the changed token cannot matter: listed with the guard
that makes it so) or a hole in the fixtures, to be closed with a new reference-generated fixture.

    python3 tools/mutate_oracle.py [--files primary,ph,...] [--jobs 6] [--sample N] [--out profiles/r4_mutation_score.json]
"""
import argparse
import concurrent.futures as cf
import json
import os
import random
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
DEFAULT_FILES = ("primary", "ph", "chem", "reset", "condenser", "sg", "init")
CHECKS = ["tests/test_oracle_golden.py tests/test_statelog_cpu.py tests/test_maintenance_cpu.py tests/test_rk4_cpu.py", "tests/test_scenarios.py"]

NUM = re.compile(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+|\.\d+(?:[eE][-+]?\d+)?)(?![\w.])")   # floating literals only
CMP = re.compile(r"(?<![<>=!\-])(<=|>=|==|!=|<|>)(?![<>=])")


def code_part(line):
    """the part of a source line that is code (no preprocessor lines, no comment text)"""
    if line.lstrip().startswith(("#", "*", "/*", "//")):
        return ""
    cut = len(line)
    for tok in ("/*", "//"):
        k = line.find(tok)
        if k >= 0:
            cut = min(cut, k)
    return line[:cut]


def sites(path):
    """[(line number, column, length, replacement, operator)] of every single-token mutation of one file"""
    out = []
    in_comment = False
    for ln, line in enumerate(open(path).read().split("\n")):
        if in_comment:
            if "*/" in line:
                in_comment = False
            continue
        code = code_part(line)
        if "/*" in line and "*/" not in line[line.find("/*"):]:
            in_comment = True
        if not code.strip() or "static_assert" in code or code.lstrip().startswith(("typedef", "struct", "}")):
            continue
        is_loop = bool(re.match(r"^\s*(for|while) \(", code))     # a loop header's comparison is a trip count, not a piece of physics: not mutated
        for m in CMP.finditer(code):
            if is_loop:
                break
            op = m.group(1)
            if op in ("<", ">") and (re.search(r"#\s*include", code) or code[m.end():m.end() + 1] == ">" or code[m.start() - 1:m.start()] == "-"):
                continue
            for rep in {"<": ("<=", ">"), ">": (">=", "<"), "<=": ("<",), ">=": (">",), "==": ("!=",), "!=": ("==",)}[op]:
                out.append((ln, m.start(1), len(op), rep, "cmp"))
        for m in re.finditer(r"(?<=[\w)\]]) ([+-]) (?=[\w(])", code):
            out.append((ln, m.start(1), 1, "-" if m.group(1) == "+" else "+", "sign"))
        for m in NUM.finditer(code):
            v = float(m.group(1))
            if v == 0.0:
                continue
            out.append((ln, m.start(1), len(m.group(1)), repr(v * (1.0 + 1e-3)), "const"))
        for m in re.finditer(r"\bnpo_py(max|min)\b", code):
            out.append((ln, m.start(0), len(m.group(0)), "npo_py" + ("min" if m.group(1) == "max" else "max"), "minmax"))
        m = re.match(r"^(\s*(?:\} else )?if \()(.*)(\)\s*(?:\{.*|[^;{]*;.*)?)$", code)
        if m and m.group(2).count("(") == m.group(2).count(")"):
            out.append((ln, len(m.group(1)), len(m.group(2)), "0 && (%s)" % m.group(2), "branch"))
            out.append((ln, len(m.group(1)), len(m.group(2)), "1 || (%s)" % m.group(2), "branch"))
    return out


def run_mutant(job):
    fname, (ln, col, length, rep, kind), idx = job
    work = tempfile.mkdtemp(prefix="npo_mut_")
    try:
        for f in os.listdir(ORACLE):
            if f.endswith((".h", ".c")):
                shutil.copy(os.path.join(ORACLE, f), work)
        path = os.path.join(work, fname)
        lines = open(path).read().split("\n")
        before = lines[ln]
        lines[ln] = before[:col] + rep + before[col + length:]
        open(path, "w").write("\n".join(lines))
        so = os.path.join(work, "libnpo.so")
        cc = subprocess.run(["gcc", "-O2", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-fopenmp", "-w",
                             "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), "-shared", "-o", so, os.path.join(work, "npo_api.c"), "-lm"],
                            capture_output=True, text=True, cwd=work)
        rec = {"file": fname, "line": ln + 1, "col": col, "op": kind, "was": before[col:col + length], "now": rep, "text": before.strip()[:140]}
        if cc.returncode != 0:
            rec["result"] = "stillborn"
            return rec
        env = dict(os.environ, NPO_LIB=so, OMP_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1")
        for check in CHECKS:
            try:
                t = subprocess.run([sys.executable, "-m", "pytest"] + check.split() + ["-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                                   capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                rec["result"] = "killed"; rec["by"] = check + " (timeout: the mutant does not terminate)"
                return rec
            if t.returncode != 0:
                m = re.search(r"FAILED (\S+)", t.stdout)
                rec["result"] = "killed"; rec["by"] = m.group(1) if m else check
                return rec
        rec["result"] = "survived"
        return rec
    finally:
        shutil.rmtree(work, ignore_errors=True)



EQUALITY = {("<", "<="), ("<=", "<"), (">", ">="), (">=", ">")}


def annotate(record):
    """Give every survivor of the record a class and, where it has one, the guard that keeps it from showing (tools/mutation_guards.py)."""
    import re
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from mutation_guards import GUARDS
    surv = record["survivors"]
    # a threshold's x 1.001 mutant counts as a sliver only if the comparison it belongs to is pinned from both sides: no dropped / forced
    # branch and no reversed comparison of the same line survives
    loose = {(r["file"], r["line"]) for r in surv if r["op"] == "branch" or (r["op"] == "cmp" and (r["was"], r["now"]) not in EQUALITY)}
    counts = {}
    for r in surv:
        why = None
        for g in GUARDS:
            f, sub, ops, text = g[:4]
            toks = g[4] if len(g) > 4 else None
            if r["file"] == f and sub in r["text"] and (ops is None or r["op"] in ops) and (toks is None or r["was"] in toks):
                why = ("guarded", text); break
        if why is None and r["op"] == "cmp" and (r["was"], r["now"]) in EQUALITY:
            why = ("equality", "differs only when the two operands are equal to the last bit")
        if why is None and r["op"] == "const" and (r["file"], r["line"]) not in loose:
            t = re.escape(r["was"])
            if re.search(r"(<=|>=|<|>)\s*\(?\s*" + t + r"(?![\w.])", r["text"]) or re.search(r"(?<![\w.])" + t + r"\s*\)?\s*(<=|>=|<|>)", r["text"]):
                why = ("sliver", "a threshold moved by 0.1 % (no dropped / forced branch and no reversed comparison of this line survives): only a value inside the 0.1 % tells")
        if why is None:
            why = ("unexplained", "")
        r["class"], r["guard"] = why
        counts[why[0]] = counts.get(why[0], 0) + 1
    record["survivor_classes"] = counts
    return counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", default=",".join(DEFAULT_FILES))
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--sample", type=int, default=0, help="mutants per file (0 = all)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r4_mutation_score.json"))
    ap.add_argument("--retest-survivors", action="store_true",
                    help="run only the mutants the record at --out lists as survivors (after fixtures were added or tolerances tightened) and move the ones now killed")
    ap.add_argument("--distinguishable-only", action="store_true",
                    help="with --retest-survivors: only the survivors tools/mutant_fuzz.py could tell from the original (the others stay listed)")
    ap.add_argument("--seed", type=int, default=4, help="of --sample")
    ap.add_argument("--only-class", default=None, help="with --retest-survivors: only the survivors --annotate put in this class (e.g. unexplained)")
    ap.add_argument("--annotate", action="store_true", help="classify the survivors of the record at --out (equality / sliver / guarded / unexplained) and list the unexplained ones")
    args = ap.parse_args()
    if args.annotate:
        rec = json.load(open(args.out))
        print(json.dumps(annotate(rec), indent=1))
        json.dump(rec, open(args.out, "w"), indent=1)
        for r in rec["survivors"]:
            if r["class"] == "unexplained":
                print("UNEXPLAINED %s:%d %s  %r -> %r   | %s   [%s]" % (r["file"], r["line"], r["op"], r["was"], r["now"], r["text"][:110], r.get("fuzz", "")[:40]))
        return
    jobs = []
    rng = random.Random(args.seed)
    if args.retest_survivors:
        old = json.load(open(args.out))
        # (records written before the column was kept match by line and token: two sites on a line with the same token are then both run)
        keep_listed = [r for r in old["survivors"] if (args.distinguishable_only and not r.get("fuzz", "").startswith("distinguishable"))
                       or (args.only_class and r.get("class") != args.only_class)]
        fuzz_of = {(r["file"], r["line"], r["op"], r["was"], r["now"]): r.get("fuzz") for r in old["survivors"]}
        want = {(r["file"], r["line"], r.get("col"), r["op"], r["was"], r["now"]) for r in old["survivors"] if r not in keep_listed}
        had = {f: c for f, c in __import__("collections").Counter(r["file"] for r in old["survivors"]).items()}
        for fname in sorted({r["file"] for r in old["survivors"]}):
            lines = open(os.path.join(ORACLE, fname)).read().split("\n")
            for i, st in enumerate(sites(os.path.join(ORACLE, fname))):
                ln, col, length, rep, kind = st
                if (fname, ln + 1, col, kind, lines[ln][col:col + length], rep) in want or (fname, ln + 1, None, kind, lines[ln][col:col + length], rep) in want:
                    jobs.append((fname, st, i))
        print("%d of %d recorded survivors found again in the present text" % (len(jobs), len(want)), flush=True)
        results = []
        with cf.ProcessPoolExecutor(args.jobs) as pool:
            for k, rec in enumerate(pool.map(run_mutant, jobs, chunksize=1)):
                results.append(rec)
                if (k + 1) % 25 == 0:
                    print("%d / %d   now killed %d" % (k + 1, len(jobs), sum(r["result"] == "killed" for r in results)), flush=True)
        for r in results:
            if fuzz_of.get((r["file"], r["line"], r["op"], r["was"], r["now"])):
                r["fuzz"] = fuzz_of[(r["file"], r["line"], r["op"], r["was"], r["now"])]
        old["survivors"] = keep_listed + [r for r in results if r["result"] == "survived"]
        left = __import__("collections").Counter(r["file"] for r in old["survivors"])
        for f, c in had.items():   # a survivor can only stay one or be killed: count by what is left
            old["by_file"][f]["killed"] += c - left.get(f, 0); old["by_file"][f]["survived"] = left.get(f, 0)
        tot = {k: sum(s_[k] for s_ in old["by_file"].values()) for k in ("killed", "survived", "stillborn")}
        old.update(checks=CHECKS, mutants=sum(tot.values()), viable=tot["killed"] + tot["survived"], killed=tot["killed"], score=tot["killed"] / max(1, tot["killed"] + tot["survived"]))
        json.dump(old, open(args.out, "w"), indent=1)
        print(json.dumps({k: v for k, v in old.items() if k != "survivors"}, indent=1))
        for r in old["survivors"]:
            print("SURVIVED %s:%d %s  %r -> %r   | %s" % (r["file"], r["line"], r["op"], r["was"], r["now"], r["text"]))
        return
    for stem in args.files.split(","):
        fname = "npo_%s.h" % stem
        ss = sites(os.path.join(ORACLE, fname))
        if args.sample and len(ss) > args.sample:
            ss = rng.sample(ss, args.sample)
        jobs += [(fname, s, i) for i, s in enumerate(ss)]
    print("%d mutants of %s" % (len(jobs), args.files), flush=True)
    results = []
    with cf.ProcessPoolExecutor(args.jobs) as pool:
        for k, rec in enumerate(pool.map(run_mutant, jobs, chunksize=1)):
            results.append(rec)
            if (k + 1) % 25 == 0:
                done = [r for r in results if r["result"] != "stillborn"]
                print("%d / %d   killed %d of %d viable" % (k + 1, len(jobs), sum(r["result"] == "killed" for r in done), len(done)), flush=True)
    summary = {}
    for r in results:
        s = summary.setdefault(r["file"], {"killed": 0, "survived": 0, "stillborn": 0})
        s[r["result"]] += 1
    viable = [r for r in results if r["result"] != "stillborn"]
    killed = sum(r["result"] == "killed" for r in viable)
    out = {"checks": CHECKS, "mutants": len(results), "viable": len(viable), "killed": killed, "score": killed / max(1, len(viable)), "by_file": summary,
           "survivors": [r for r in results if r["result"] == "survived"]}
    if os.path.exists(args.out):      # a partial run over some files updates the record of the others
        try:
            old = json.load(open(args.out))
            keep = [r for r in old.get("survivors", []) if r["file"] not in summary]
            for f, s in old.get("by_file", {}).items():
                out["by_file"].setdefault(f, s)
            out["survivors"] = keep + out["survivors"]
            tot = {k: sum(s[k] for s in out["by_file"].values()) for k in ("killed", "survived", "stillborn")}
            out.update(mutants=sum(tot.values()), viable=tot["killed"] + tot["survived"], killed=tot["killed"], score=tot["killed"] / max(1, tot["killed"] + tot["survived"]))
        except Exception:
            pass
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "survivors"}, indent=1))
    for r in out["survivors"]:
        print("SURVIVED %s:%d %s  %r -> %r   | %s" % (r["file"], r["line"], r["op"], r["was"], r["now"], r["text"]))


if __name__ == "__main__":
    main()
