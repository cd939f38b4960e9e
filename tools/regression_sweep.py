#!/usr/bin/env python3
"""The reference's own serial D3Q19 regression suite (tests/regression/
d3q19-short: 112 inputs, each with the log its authors keep) run through the
reference's executable built for gfx950 WITH the binding, and -- to tell what
the binding changed from what the reference's HIP target changes by itself --
through the same executable without it.

  collect   (development container; needs /root/reference) stage the inputs and
            their logs of the whole suite under tools/_sweep_data/ -- data the
            reference's tests hold, no source; git-ignored, it travels to the
            GPU box with the gpurun snapshot like the built binaries. (The nine
            cases a committed TEST reads are kept under
            tests/golden/regression_d3q19_short/.)
  run       (GPU box) run every input, compare each log with the expected one
            line by line: same words, numbers within --tol (the reference's
            tests/awk-fp-diff.sh uses 1e-12 on the printed values), after
            dropping the lines its tests/test-diff.sh drops (timers, version,
            compiler, target) and the binding's own "liblbmi:" lines

  report    one table from the .jsonl files of `run`

usage: regression_sweep.py collect
       regression_sweep.py run [--first I --last J] [--only NAME,...]
                               [--unbound 1] [--mode halo] [--env K=V,...] [--out FILE]
"""

import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# the cases committed tests read; the whole suite when `collect` has staged it
KEPT = os.path.join(ROOT, "tests", "golden", "regression_d3q19_short")
STAGED = os.path.join(HERE, "_sweep_data")
DATA = STAGED if os.path.isdir(STAGED) else KEPT
REFDIR = "/root/reference/tests/regression/d3q19-short"
EXE = os.path.join(ROOT, "oracle", "_ref", "ludwig_hip_d3q19")

# lines that cannot agree between two builds (what tests/test-diff.sh removes)
DROP = re.compile(
    r"call\)|calls\)|Welcome|Git commit:|Compiler:|^\s*name:|^\s*version-string:"
    r"|^\s*options:|Target thread model:|Default threads per block|OpenMP|^Halo type:"
    r"|Note assertions|SVN.revision|^\s*$|Timer|user.parameters.from|GPU INFO"
    r"|SIMD vector|Start time|End time|^liblbmi:")
NUM = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$")

# the reference's OWN HIP target ends in a GPU memory fault on these (first
# sweep, unbound executable; liquid crystal + colloids): never run again
REF_FAULTS = {"serial-chol-n01", "serial-chol-n02", "serial-chol-n03", "serial-chol-n04",
              "serial-chol-p01", "serial-chol-st1", "serial-chol-st2", "serial-chol-st7"}

# inputs that need something the suite's Makefile prepares first
NEEDS = {"serial-rest-c02": "serial-rest-c01",       # restart from c01's files
         "iodrop-mpi1-io3": "iodrop-mpi1-io2",       # restart from io2's files
         "serial-poly-st1": "util/multi_poly_init"}  # a generated initial state


def normalise(text):
    out = []
    for line in text.splitlines():
        if DROP.search(line):
            continue
        line = re.sub(r"d3q19 R", "d3q19", line)
        out.append(line.split())
    return out


def same_line(la, lb_, tol):
    """-> (bool, largest numeric difference)"""
    if len(la) != len(lb_):
        return False, 0.0
    ok, worst = True, 0.0
    for x, y in zip(la, lb_):
        xs, ys = x.rstrip(",;:"), y.rstrip(",;:")
        if NUM.match(xs) and NUM.match(ys):
            d = abs(float(xs) - float(ys))
            worst = max(worst, d)
            if not d < tol:
                ok = False
        elif x != y:
            ok = False
    return ok, worst


def compare(expected, got, tol):
    """-> (number of lines that differ, largest numeric difference of lines
    that pair up, first differing pair). Lines are paired by difflib on their
    words with numbers masked, so an extra line does not shift the rest."""
    import difflib
    a, b = normalise(expected), normalise(got)
    key = lambda l: " ".join("#" if NUM.match(t.rstrip(",;:")) else t for t in l)
    sm = difflib.SequenceMatcher(None, [key(l) for l in a], [key(l) for l in b], autojunk=False)
    bad, worst, first = 0, 0.0, None
    for tag, i0, i1, j0, j1 in sm.get_opcodes():
        if tag == "equal":
            for la, lb_ in zip(a[i0:i1], b[j0:j1]):
                ok, w = same_line(la, lb_, tol)
                worst = max(worst, w)
                if not ok:
                    bad += 1
                    if first is None:
                        first = (" ".join(la), " ".join(lb_))
        else:
            bad += max(i1 - i0, j1 - j0)
            if first is None:
                first = (" / ".join(" ".join(l) for l in a[i0:i1])[:300],
                         " / ".join(" ".join(l) for l in b[j0:j1])[:300])
    return bad, worst, first


def names():
    return sorted(f[:-4] for f in os.listdir(DATA) if f.endswith(".inp"))


def collect():
    global DATA
    DATA = STAGED
    os.makedirs(DATA, exist_ok=True)
    n = 0
    # d3q19-short: every serial input; d3q19-io: the one-rank runs of the
    # droplet with the three i/o arrangements (one file, two, ASCII records)
    for d, prefix in ((REFDIR, "serial-"), (REFDIR.replace("d3q19-short", "d3q19-io"), "iodrop-mpi1-")):
        for f in sorted(os.listdir(d)):
            if f.startswith(prefix) and (f.endswith(".inp") or f.endswith(".log")):
                shutil.copyfile(os.path.join(d, f), os.path.join(DATA, f))
                n += 1
    print("%d files -> %s" % (n, DATA))


def run_one(name, exe, env, workdir, limit):
    dst = os.path.join(workdir, "input")
    if os.path.exists(dst):
        os.remove(dst)                     # (a restart chain shares its directory)
    src = os.path.join(DATA, name + ".inp")
    if not os.path.exists(src):
        src = os.path.join(KEPT, name + ".inp")     # a case the tests keep
    shutil.copyfile(src, dst)
    t0 = time.time()
    try:
        r = subprocess.run([exe], cwd=workdir, env=env, capture_output=True,
                           text=True, timeout=limit)
        return r.returncode, r.stdout, r.stderr, time.time() - t0
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        return -999, out, "timeout after %d s" % limit, time.time() - t0


def run(args):
    todo = names()
    if args.only:
        pick = args.only.split(",")
        todo = [n for n in todo if n in pick or n[7:] in pick or n[7:11] in pick]
    else:
        todo = todo[args.first:args.last]
    env = dict(os.environ)
    for k in ("LBMI_MODE", "LBMI_FE", "LBMI_HYDRO"):
        env.pop(k, None)
    if args.mode:
        env["LBMI_MODE"] = args.mode
    for kv in filter(None, args.env.split(",")):
        k, v = kv.split("=")
        env[k] = v
    # the unbound executable first: an input it faults on is not given to the
    # bound one (a GPU fault is evidence enough once)
    exe_ = EXE.replace("d3q19", "d3q%d" % args.nvel)
    # (--nvel 27: the suite's inputs with the D3Q27 build; the kept logs are
    # D3Q19 runs, so only bound against unbound means anything there)
    legs = ([("unbound", exe_)] if args.unbound else []) + [("bound", exe_ + "_shim")]
    for _, exe in legs:
        if not os.path.exists(exe):
            raise SystemExit(exe + " is missing: `make -C oracle hip` in the development container")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    keep = os.path.splitext(args.out)[0] + "_logs"
    os.makedirs(keep, exist_ok=True)
    res = []
    with open(args.out, "a") as fh:
        # restart chains share a directory; everything else gets a fresh one
        shared = {}
        for name in todo:
            expected = open(os.path.join(DATA, name + ".log")).read()
            rec = {"name": name}
            outputs = {}
            for leg, exe in legs:
                if NEEDS.get(name, "").startswith("util/"):
                    rec[leg] = {"status": "not run", "why": "needs " + NEEDS[name]}
                    continue
                if name in REF_FAULTS:
                    rec[leg] = {"status": "not run", "why": "the reference's HIP target faults on it"}
                    continue
                if leg == "bound" and rec.get("unbound", {}).get("rc", 0) < 0:
                    rec[leg] = {"status": "not run", "why": "the unbound executable was killed by a signal"}
                    continue
                chain = name if name in NEEDS.values() else NEEDS.get(name)
                if chain:
                    wd = shared.setdefault((leg, chain), tempfile.mkdtemp())
                    tmp = None
                else:
                    tmp = tempfile.TemporaryDirectory()
                    wd = tmp.name
                rc, out, err, dt = run_one(name, exe, dict(env, LBMI_REPORT="1"), wd, args.limit)
                if tmp:
                    tmp.cleanup()
                done = "Ludwig finished normally."
                if rc != 0 or (done in expected and done not in out):
                    rec[leg] = {"status": "did not finish", "rc": rc, "seconds": round(dt, 1),
                                "tail": (out[-400:] + " | " + err[-400:])}
                    with open(os.path.join(keep, "%s_%s.log" % (name, leg)), "w") as lf:
                        lf.write(out + "\n--- stderr ---\n" + err)
                    continue
                bad, worst, first = compare(expected, out, args.tol)
                rec[leg] = {"status": "same" if bad == 0 else "differs", "lines": bad,
                            "worst": worst, "seconds": round(dt, 1)}
                outputs[leg] = out
                if leg == "bound":
                    # LBMI_REPORT=1 (integration/ludwig_shim.c): calls the library
                    # took / calls it handed to the original, per bound symbol
                    calls = {}
                    for line in err.splitlines():
                        w = line.split()
                        if line.startswith("liblbmi report:") and len(w) == 5 and w[3].isdigit():
                            calls[w[2]] = [int(w[3]), int(w[4])]
                    rec["calls"] = calls
                    # ... and how the binding ended up running it: "execution
                    # mode fused (chosen by the binding); rho, u on demand in
                    # 10 of 10 collisions"
                    m = re.search(r"liblbmi report: execution mode (\w+) \(([^)]*)\); rho, u on "
                                  r"demand in (\d+) of (\d+) collisions", err)
                    if m:
                        rec["policy"] = {"mode": m.group(1), "lazy": int(m.group(3)),
                                         "collisions": int(m.group(4))}
                if leg == "bound" and "unbound" in outputs:
                    # what the binding changes: the two executables on this GPU
                    b2, w2, f2 = compare(outputs["unbound"], out, args.tol)
                    rec["bound_vs_unbound"] = {"lines": b2, "worst": w2}
                    if b2:
                        rec["bound_vs_unbound"]["first"] = f2
                if bad:
                    rec[leg]["first"] = first
                    with open(os.path.join(keep, "%s_%s.log" % (name, leg)), "w") as lf:
                        lf.write(out)
            fh.write(json.dumps(rec) + "\n")
            fh.flush()
            print(name, " ".join("%s=%s(%.1fs)" % (l, rec[l]["status"], rec[l].get("seconds", 0))
                                 for l, _ in legs), flush=True)
            res.append(rec)
        for d in shared.values():
            shutil.rmtree(d, ignore_errors=True)
    for leg, _ in legs:
        tally = {}
        for r in res:
            tally[r[leg]["status"]] = tally.get(r[leg]["status"], 0) + 1
        print(leg, tally)
    both = [r for r in res if "bound_vs_unbound" in r]
    print("bound against unbound: %d of %d identical within %g" % (
        sum(1 for r in both if r["bound_vs_unbound"]["lines"] == 0), len(both), args.tol))


def report(args):
    """One line per input from the .jsonl files of `run`."""
    rows = {}
    for f in args.files:
        for line in open(f):
            r = json.loads(line)
            rows[r["name"]] = r
    def cell(x):
        if x is None:
            return "-"
        if x["status"] in ("same", "differs"):
            return "%d lines, %.1e" % (x["lines"], x["worst"])
        if x["status"] == "not run":
            return "not run: " + x["why"]
        return "did not finish (rc %s)" % x["rc"]
    print("# %-10s | %-28s | %-28s | %-22s | %s" % (
        "input", "unbound vs the kept log", "bound vs the kept log", "bound vs unbound",
        "mode at exit, collisions with rho,u on demand | library/original calls: collide halo propagation | others the library took"))
    n_same = n_both = 0
    for name in sorted(rows):
        r = rows[name]
        bu = r.get("bound_vs_unbound")
        if bu is not None:
            n_both += 1
            n_same += (bu["lines"] == 0)
        c = r.get("calls") or {}
        main = " ".join("%d/%d" % tuple(c.get(k, [0, 0])) for k in ("lb_collide", "lb_halo_swap", "lb_propagation"))
        rest = ",".join(k for k in sorted(c) if k not in ("lb_collide", "lb_halo_swap", "lb_propagation") and c[k][0])
        pol = r.get("policy")
        if pol:
            main = "%s, rho,u on demand %d/%d | %s" % (pol["mode"], pol["lazy"], pol["collisions"], main)
        print("%-12s | %-28s | %-28s | %-22s | %s" % (
            name[7:], cell(r.get("unbound")), cell(r.get("bound")),
            "-" if bu is None else ("identical, %.1e" % bu["worst"] if bu["lines"] == 0
                                    else "%d lines, %.1e" % (bu["lines"], bu["worst"])),
            (main + " | " + rest) if c else "-"))
    print("# bound against unbound: %d of %d logs identical (numbers within the tolerance)" % (n_same, n_both))
    pols = [r["policy"] for r in rows.values() if r.get("policy")]
    if pols:
        print("# of the %d runs that reached a bound lb_collide: %d ended in fused, %d in halo, %d in eager; "
              "%d left rho, u on demand in every collision, %d in none"
              % (len(pols), sum(p["mode"] == "fused" for p in pols),
                 sum(p["mode"] == "halo" for p in pols), sum(p["mode"] == "eager" for p in pols),
                 sum(p["lazy"] == p["collisions"] and p["collisions"] > 0 for p in pols),
                 sum(p["lazy"] == 0 for p in pols)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["collect", "run", "report"])
    ap.add_argument("files", nargs="*")
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--last", type=int, default=None)
    ap.add_argument("--only", default="")
    ap.add_argument("--unbound", type=int, default=1)
    ap.add_argument("--nvel", type=int, default=19, choices=[19, 27])
    ap.add_argument("--mode", default="")
    ap.add_argument("--env", default="", help="KEY=VALUE,... for the runs (LBMI_FE=1,LBMI_HYDRO=lazy)")
    ap.add_argument("--tol", type=float, default=1e-12)
    ap.add_argument("--limit", type=int, default=240)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "regression_sweep.jsonl"))
    args = ap.parse_args()
    if args.what == "collect":
        collect()
    elif args.what == "report":
        report(args)
    else:
        run(args)


if __name__ == "__main__":
    sys.exit(main())
