#!/usr/bin/env python3
"""Do the GPU parity tests reject a wrong DEVICE text?  A sample of tools/mutate_oracle.py's mutants, made in the HIP headers instead.

tools/mutate_oracle.py measures how much of the CPU restatement the reference's fixtures pin.  What ships is the device code
(nuclear_sim_amd/csrc/npd_*.h), which the `-m gpu` tests hold to the same fixtures.  This tool makes the same single-token mutants in
the device headers of the subsystems whose text the two share (primary, ph, chem, reset, condenser, sg, init), a seeded sample of them,
and builds each into its own libnpb.so (only npb_kernels.hip's fp64 object is recompiled; the rest is linked from nuclear_sim_amd/build):

    python3 tools/mutate_device.py build [--sample 24] [--jobs 6]      here (hipcc cross-compiles): tools/device_mutants/libnpb_<k>.so + index.json
    python3 tools/mutate_device.py run                                 on the GPU box: the golden replays against each, verdicts into
                                                                        gpurun_out/r4/device_mutants.json

Each sampled site is looked up in the restatement (same line after npd_ -> npo_): a mutant whose twin the CPU fixtures KILL is expected to
be killed on the GPU too; one whose twin is a listed survivor (profiles/r4_mutation_score.json: equality / sliver / guarded) is expected to
survive.  Anything else is a finding.
"""
import argparse
import concurrent.futures as cf
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mutate_oracle as M   # noqa: E402

CSRC = os.path.join(ROOT, "nuclear_sim_amd", "csrc")
BUILD = os.path.join(ROOT, "nuclear_sim_amd", "build")
OUT = os.path.join(ROOT, "tools", "device_mutants")
STEMS = ("primary", "ph", "chem", "reset", "condenser", "sg", "init")
HIPFLAGS = "-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -freciprocal-math -fapprox-func -fvisibility=hidden -w".split()


def device_sites(stem):
    """mutation sites of a device header, with the restatement's spelling of the clip helpers understood"""
    path = os.path.join(CSRC, "npd_%s.h" % stem)
    tmp = tempfile.NamedTemporaryFile("w", suffix=".h", delete=False)
    tmp.write(open(path).read().replace("npd_py", "npo_py")); tmp.close()     # same length: columns stay valid
    try:
        ss = M.sites(tmp.name)
    finally:
        os.unlink(tmp.name)
    lines = open(path).read().split("\n")
    # not on the GPU: a mutant of a line that computes an INDEX (ejector rotation, pump count) could address outside a private array,
    # and a faulting kernel can take the whole host down -- those lines are left to the CPU run
    risky = ("lead_ejector", "lag_ejector", "lead_index", "lag_index", "needed", "pumps_needed", "remaining_stages", "NPD_EXT_IDX", "idx", "index")

    def inside_brackets(line, col):
        depth = 0
        for ch in line[:col]:
            depth += (ch == "[") - (ch == "]")
        return depth > 0
    return [(ln, col, length, rep.replace("npo_py", "npd_py"), kind) for ln, col, length, rep, kind in ss
            if not any(w in lines[ln] for w in risky) and not inside_brackets(lines[ln], col)]


def oracle_twin(stem, text, op, was, now, survivors):
    """what the CPU fixtures did with the same mutation of the same line of the restatement"""
    want = text.replace("npd_", "npo_").replace("NPD_", "NPO_").replace("__device__ __forceinline__", "NPO_FN").strip()
    if not os.path.exists(os.path.join(ROOT, "oracle", "npo_%s.h" % stem)):
        return "no twin line"
    lines = [l.strip() for l in open(os.path.join(ROOT, "oracle", "npo_%s.h" % stem)).read().split("\n")]
    if want not in lines:
        return "no twin line"
    for r in survivors:
        if r["file"] == "npo_%s.h" % stem and r["text"] == want[:140] and r["op"] == op and r["was"] == was.replace("npd_", "npo_") and r["now"] == now.replace("npd_", "npo_"):
            return "survivor (%s)" % r.get("class", "listed")
    return "killed"


def build_one(job):
    k, stem, (ln, col, length, rep, kind) = job
    work = tempfile.mkdtemp(prefix="npd_mut_")
    try:
        os.makedirs(os.path.join(work, "nuclear_sim_amd"))
        shutil.copytree(CSRC, os.path.join(work, "nuclear_sim_amd", "csrc"))       # the headers include ../../include/ by relative path
        os.symlink(os.path.join(ROOT, "include"), os.path.join(work, "include"))
        path = os.path.join(work, "nuclear_sim_amd", "csrc", "npd_%s.h" % stem)
        lines = open(path).read().split("\n")
        before = lines[ln]
        lines[ln] = before[:col] + rep + before[col + length:]
        open(path, "w").write("\n".join(lines))
        obj = os.path.join(work, "k64.o")
        cc = subprocess.run(["/opt/rocm/bin/hipcc"] + HIPFLAGS + ["-I", os.path.join(ROOT, "include"), "-c", "-o", obj, os.path.join(work, "nuclear_sim_amd", "csrc", "npb_kernels.hip")],
                            capture_output=True, text=True)
        rec = {"k": k, "file": "npd_%s.h" % stem, "line": ln + 1, "op": kind, "was": before[col:col + length], "now": rep, "text": before.strip()[:140]}
        if cc.returncode != 0:
            rec["build"] = "stillborn"
            return rec
        so = os.path.join(OUT, "libnpb_%d.so" % k)
        ld = subprocess.run(["/opt/rocm/bin/hipcc"] + HIPFLAGS + ["-shared", "-pthread", "-o", so, obj, os.path.join(BUILD, "npb_kernels_f32.o"),
                                                                 os.path.join(BUILD, "npb_api_f64.o"), os.path.join(BUILD, "npb_seeds.o")], capture_output=True, text=True)
        rec["build"] = "ok" if ld.returncode == 0 else "link failed"
        return rec
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cmd_build(args):
    for f in ("npb_kernels_f32.o", "npb_api_f64.o", "npb_seeds.o"):
        if not os.path.exists(os.path.join(BUILD, f)):
            sys.exit("build the product library first (make -C nuclear_sim_amd/csrc): %s is missing" % f)
    shutil.rmtree(OUT, ignore_errors=True); os.makedirs(OUT)
    rng = random.Random(args.seed)
    stems = tuple(args.stems.split(",")) if args.stems else STEMS
    pool_sites = [(stem, s) for stem in stems for s in device_sites(stem)]
    picks = rng.sample(pool_sites, args.sample)
    record = json.load(open(os.path.join(ROOT, "profiles", "r4_mutation_score.json")))
    jobs = [(k, stem, s) for k, (stem, s) in enumerate(picks)]
    out = []
    with cf.ProcessPoolExecutor(args.jobs) as pool:
        for rec in pool.map(build_one, jobs):
            rec["twin"] = oracle_twin(rec["file"][4:-2], rec["text"], rec["op"], rec["was"], rec["now"], record["survivors"])
            out.append(rec)
            print("%2d %-16s %4d %-6s %-14r -> %-22r build %-9s twin %s" % (rec["k"], rec["file"], rec["line"], rec["op"], rec["was"], rec["now"][:22], rec["build"], rec["twin"]), flush=True)
    json.dump(out, open(os.path.join(OUT, "index.json"), "w"), indent=1)


def cmd_run(args):
    index = json.load(open(os.path.join(OUT, "index.json")))
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r4"), exist_ok=True)
    if args.stage2:
        # the mutants the golden replays let through although the CPU fixtures kill their twin: the whole GPU parity file (the HIP-vs-restatement
        # comparisons on random batches, the rk4 mode, the long run ...)
        index = json.load(open(args.stage2))
        for rec in index:
            if rec.get("gpu") != "survived" or (rec["twin"] != "killed" and not args.stage2_all):
                continue
            env = dict(os.environ, NPB_LIB=os.path.join(OUT, "libnpb_%d.so" % rec["k"]), PYTHONDONTWRITEBYTECODE="1")
            t = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                                "-k", "not test_native_library_is_loaded"], cwd=ROOT, env=env, capture_output=True, text=True)
            import re
            m = re.search(r"FAILED (\S+)", t.stdout)
            rec["gpu_stage2"] = "survived" if t.returncode == 0 else "killed"
            rec["by_stage2"] = (m.group(1) if m else t.stdout[-200:]) if t.returncode != 0 else ""
            print("%2d %-16s %4d %-6s stage 2: %-8s %s" % (rec["k"], rec["file"], rec["line"], rec["op"], rec["gpu_stage2"], rec["by_stage2"]), flush=True)
            json.dump(index, open(args.stage2, "w"), indent=1)
        return
    for rec in index:
        if rec["build"] != "ok":
            continue
        env = dict(os.environ, NPB_LIB=os.path.join(OUT, "libnpb_%d.so" % rec["k"]), PYTHONDONTWRITEBYTECODE="1")
        try:
            t = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                                "-k", "test_hip_replays_golden and not every_step_kernel"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
            rec["gpu"] = "survived" if t.returncode == 0 else "killed"
            if t.returncode != 0:
                import re
                m = re.search(r"FAILED (\S+)", t.stdout)
                rec["by"] = m.group(1) if m else t.stdout[-200:]
        except subprocess.TimeoutExpired:
            rec["gpu"] = "killed"; rec["by"] = "timeout"
        print("%2d %-16s %4d %-6s twin %-22s gpu %-8s %s" % (rec["k"], rec["file"], rec["line"], rec["op"], rec["twin"], rec["gpu"], rec.get("by", "")), flush=True)
        json.dump(index, open(os.path.join(ROOT, "gpurun_out", "r4", "device_mutants.json"), "w"), indent=1)
    ok = [r for r in index if r.get("gpu")]
    exp_k = [r for r in ok if r["twin"] == "killed"]
    print("twin killed on the CPU: %d, of them killed on the GPU: %d;  twin a listed survivor: %d, of them surviving on the GPU: %d;  no twin: %d (killed %d)" % (
        len(exp_k), sum(r["gpu"] == "killed" for r in exp_k),
        sum(r["twin"].startswith("survivor") for r in ok), sum(r["twin"].startswith("survivor") and r["gpu"] == "survived" for r in ok),
        sum(r["twin"] == "no twin line" for r in ok), sum(r["twin"] == "no twin line" and r["gpu"] == "killed" for r in ok)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("--sample", type=int, default=24)
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--stems", default=None, help="build: device headers to mutate (default: the seven whose text the restatement shares), e.g. feedwater,turbine,lube")
    ap.add_argument("--stage2-all", action="store_true", help="with --stage2: every survivor of the replays, whatever its twin")
    ap.add_argument("--stage2", default=None, help="run: a results file of an earlier `run`; its survivors with a CPU-killed twin get the whole GPU parity file")
    a = ap.parse_args()
    (cmd_build if a.cmd == "build" else cmd_run)(a)
