#!/usr/bin/env python3
"""MLUPS benchmark of the LB time step (lb_collide + lb_halo + lb_propagation)
on D3Q19 256^3 (BASELINE.json), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over the whole lattice through the
C-ABI: lbmi_lb_collide, lbmi_lb_halo, lbmi_lb_propagation. In the default
FUSED mode the three calls execute as ONE kernel (pull-propagation with
periodic index wrap + collision); --mode eager runs the reference's three
separate stages. Inputs are resident in HBM before the timed region.

The hydro arrays lb_collide borrows (force, rho, u) are there in every
default run. --hydro lazy (default): the force field has been zeroed through
the library (hydro_f_zero), which therefore does not read it, and rho, u of
the last collision are formed when asked for (lbmi_lb_hydro_sync, INSIDE the
timed region, once, as a run that prints statistics at its end would);
--hydro 1: read and stored by every collision, as the reference does. The
second is also measured in every default run, after the timed region, and
reported under "hydro_every_step".

N > 1: `python bench.py --gpus N` starts its own N ranks (torch.distributed.run
on 127.0.0.1, before anything touches a GPU); under an existing launcher it
just takes its rank. Strong scaling by default: the 256^3 box is cut into N
slabs along X (the reference's `grid N_1_1`), X halo planes travel over RCCL
(ncclSend/ncclRecv) on a second stream overlapped with the interior planes.
--config 5 is the weak-scaling D3Q27 case (64x512x256 per GPU).

Prints ONE JSON line on rank 0. "roofline" follows SURVEY.md 8(d): achieved =
2*Q*8 bytes per lattice update (304 B for D3Q19) over the average launch time
of the step's kernel (HIP events around sampled launches, on the library's
stream). At N = 1 two reported comparators are measured after the timed
region: "cpu_baseline" (the reference's CPU build, oracle/_ref, on this host's
cores) and "reference_gpu" (the reference's own HIP back end on this GPU).
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Before anything initialises HIP: the host driver supports dmabuf IPC only
# (RCCL / device-memory sharing across processes fails without this), and all
# ranks of a run share one node, so the control plane (gloo) and the RCCL
# bootstrap go over loopback whatever the host name resolves to.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

PRELOAD_STEPS = 3
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--nvel", type=int, default=19)
    ap.add_argument("--scheme", default="m10", choices=["m10", "bgk", "trt"])
    ap.add_argument("--mode", default="fused",
                    choices=["fused", "eager", "inplace", "fused_soa", "fused_halo"],
                    help="fused (default): on one GPU the deferred state is "
                    "kept in the block-contiguous order; fused_soa: in the "
                    "reference's SoA order")
    ap.add_argument("--hydro", default="lazy", choices=["lazy", "1", "0"],
                    help="lazy: hydro arrays present, force zeroed through "
                    "the library (not read), rho,u on demand; 1: lb_collide "
                    "reads hydro->force and writes hydro->rho,u every step "
                    "as the reference does; 0: NULL hydro arrays")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="a BASELINE.json configuration (1-based): 1 = D3Q19 "
                    "BGK 64^3; 2, 3 = D3Q19 M10 256^3 (the default; 3 with "
                    "--gpus 8); 4 = D3Q19 + symmetric free energy 128^3; "
                    "5 = D3Q27 M10 64x512x256 per GPU, weak scaling")
    ap.add_argument("--dry-run", type=int, default=0,
                    help="1: ranks, decomposition and the X exchange schedule "
                    "only (gloo on the CPU, no device call)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="weak: --size is the per-GPU box, stacked along X")
    ap.add_argument("--nhalo", type=int, default=1)
    ap.add_argument("--force-field", type=int, default=0,
                    help="1: per-site force F = 1e-5 cos(2 pi x/L) (BASELINE config 4)")
    ap.add_argument("--fe", default="none", choices=["none", "symmetric"],
                    help="symmetric: the full binary-fluid step of BASELINE "
                    "config 4 (phi halo, thermodynamic force, Cahn-Hilliard, "
                    "LB step); 1 GPU, forces --nhalo 2")
    ap.add_argument("--fe-grad", type=int, default=7, choices=[7, 27],
                    help="fd_gradient_calculation 3d_7pt_fluid | 3d_27pt_fluid")
    ap.add_argument("--fe-order", type=int, default=1, choices=[1, 2, 3, 4],
                    help="fd_advection_scheme_order")
    ap.add_argument("--fe-route", default="step", choices=["step", "phi", "grad"],
                    help="step: lbmi_symmetric_lb_step, the whole coupled step "
                    "in one call (one kernel in the steady state: the force "
                    "never leaves the registers; needs 7-point gradients, one "
                    "GPU; falls back to phi otherwise); phi: force + "
                    "Cahn-Hilliard in one pass straight from phi, then the LB "
                    "step; grad: gradient arrays first (lbmi_field_grad), then "
                    "the one pass reading them")
    ap.add_argument("--fe-rho", default="lazy", choices=["lazy", "store"],
                    help="--fe-route step: hydro->rho on demand (once, inside "
                    "the timed region; the binding's default with a free "
                    "energy) or stored by every step")
    ap.add_argument("--fe-halos", type=int, default=0,
                    help="1: keep the field halo swaps of phi and u (the "
                    "reference's structure) on one GPU as well")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "peer"],
                    help="N > 1: rccl = one process per GPU, ncclSend/ncclRecv "
                    "(the driver's launch); peer = ONE process, one host thread "
                    "per GPU, the planes as peer-to-peer copies over xGMI "
                    "(lbmi_ring_t; no RCCL bootstrap; ranks beyond the number "
                    "of devices share them: a rehearsal). `python bench.py "
                    "--gpus N --transport peer` from a plain shell")
    ap.add_argument("--cartdim", type=int, default=0, choices=[0, 1, 2],
                    help="with --selfring 1: the direction of the 1-rank ring "
                    "(0 X: contiguous planes, interior + boundary launch; 1 Y, "
                    "2 Z: gathered planes, one launch + the face launch)")
    ap.add_argument("--selfring", type=int, default=0,
                    help="1 GPU only: route the X halo through a 1-rank RCCL "
                    "ring (exercises the N>1 step path: pack, send/recv, "
                    "unpack, interior/boundary split)")
    ap.add_argument("--noise", type=float, default=0.0,
                    help="kT > 0: isothermal fluctuations (lbmi_noise_set; D3Q19, "
                    "--mode eager or fused_halo): +32 B/site of generator state per step")
    ap.add_argument("--tune", default="", help="key=value,... (lbmi_tune)")
    ap.add_argument("--timing-period", type=int, default=0,
                    help="HIP-event timing of every k-th kernel launch inside "
                    "the timed region (1 = all; 0 = about 25 samples, 6 for slabs)")
    ap.add_argument("--own-stream", type=int, default=0,
                    help="1: the library's private non-blocking stream "
                    "instead of torch's current (the legacy default) stream")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=8)
    args = ap.parse_args()
    if args.config == 1:
        args.size, args.scheme = [64, 64, 64], "bgk"
    if args.config == 4:
        args.size, args.fe, args.nhalo, args.hydro = [128, 128, 128], "symmetric", 2, "1"
    if args.config == 5:
        args.size, args.nvel, args.scaling = [64, 512, 256], 27, "weak"
    if args.fe != "none" and args.hydro != "1":
        args.hydro = "1"             # u is read by every step of the free-energy pass
    if args.timing_period <= 0:
        args.timing_period = max(1, min(8, args.steps // 24))
        if args.gpus > 1 or args.selfring or int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # a sampled slab step carries eight event records on three
            # streams (2.6 % on the step at every 8th, nothing measurable at
            # every 32nd): six samples are enough for the per-rank phases
            args.timing_period = max(1, args.steps // 6)
    return args


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell: start N ranks under
    torch.distributed.run on the loopback interface and pass their one JSON
    line on. Called before torch is imported, so this process never touches a
    GPU; the ranks are children, not replacements of it."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    # the ranks are fresh child processes (never a re-exec of this one); a
    # rank that hangs must not hang the caller: the whole group is ended and
    # the exit code says so
    limit = float(os.environ.get("LBMI_BENCH_TIMEOUT", "1500"))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, _ = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(p.pid, signal.SIGKILL)       # exactly the group started here
        p.wait()
        sys.stderr.write("bench.py: the %d ranks did not finish within %.0f s\n"
                         % (args.gpus, limit))
        raise SystemExit(124)
    r = subprocess.CompletedProcess(cmd, p.returncode, out)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout)
        raise SystemExit(r.returncode or 1)
    return lines[-1]


def cpu_baseline(args):
    """Reference (oracle/_ref, kind 'reference') or oracle port (kind 'port')
    timed on this host's cores: same box, same scheme, a few steps."""
    cores = len(os.sched_getaffinity(0))
    exe = os.path.join(ROOT, "oracle", "_ref",
                       "ref_driver_d3q%d_fast" % args.nvel)
    size = list(args.size)
    zeta = 0.3 if args.scheme == "m10" else 0.1
    sample = "%dx%dx%d %s, %d steps after 1 warm-up" % (*size, args.scheme,
                                                         args.cpu_steps)
    if os.path.exists(exe):
        # The GPU box gives a process a CPU share of ~16 cores although it
        # sees all 256 hardware threads; the reference's OpenMP kernels
        # collapse when oversubscribed (8 MLUPS at 256 threads against 53-56
        # at 16-32, profiles/r01_cpu_baseline_threads.txt). Time it at 16 and
        # at 32 threads and report the better one.
        best = None
        for nthr in sorted({min(16, cores), min(32, cores)}):
            env = dict(os.environ, OMP_NUM_THREADS=str(nthr))
            try:
                out = subprocess.run(
                    [exe, "time", *map(str, size), args.scheme, "0.1",
                     repr(zeta), str(args.cpu_steps)], env=env, check=True,
                    capture_output=True, text=True, timeout=900).stdout
                r = json.loads(out.strip().splitlines()[-1])
                if best is None or r["mlups"] > best["mlups"]:
                    best = r
            except Exception as e:      # fall through to the port
                sys.stderr.write("cpu_baseline: reference run failed: %r\n" % e)
        if best is not None:
            r = best
            return {"value": round(r["mlups"], 3), "unit": "MLUPS",
                    "cores": r["threads"], "kind": "reference",
                    "sample": sample + " (the reference itself, oracle/_ref: "
                    "gcc -O2 -DNDEBUG -fopenmp, its default AoS order, "
                    "lb_collide+lb_halo+lb_propagation; best of 16/32 OpenMP "
                    "threads; t_collide/halo/prop = %.3f/%.3f/%.3f s)"
                    % (r["t_collide"], r["t_halo"], r["t_propagation"])}
    import numpy as np
    from oracle import lb_oracle as lbo
    os.environ["OMP_NUM_THREADS"] = str(cores)
    p = lbo.make_param(args.nvel, size, 1, args.scheme, 0.1, zeta)
    f = lbo.init_synthetic(p)
    fp = np.zeros_like(f)
    f, fp = lbo.step(p, f, fp)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        f, fp = lbo.step(p, f, fp)
    dt = time.perf_counter() - t0
    mlups = 1e-6 * size[0] * size[1] * size[2] * args.cpu_steps / dt
    return {"value": round(mlups, 3), "unit": "MLUPS", "cores": cores,
            "kind": "port", "sample": sample + " (oracle/lb_oracle.c, OpenMP)"}


def reference_gpu(args):
    """The reference's OWN HIP back end (its TargetDP kernels, built for gfx950
    by `make -C oracle hip` into oracle/_ref) on this GPU, same lattice, same
    three calls per step: a reported comparator like cpu_baseline, measured
    after the timed region in a child process. None if the binary is absent or
    the scheme is not M10 (on a device the reference relaxes every scheme as
    M10, tests/test_gpu_shim.py)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver_hip_d3q%d" % args.nvel)
    if not os.path.exists(exe) or args.scheme != "m10" or args.fe != "none":
        return None
    try:
        out = subprocess.run([exe, "time", *map(str, args.size), "m10", "0.1",
                              "0.3", "10"], check=True, capture_output=True,
                             text=True, timeout=300).stdout
        r = json.loads(out.strip().splitlines()[-1])
    except Exception as e:
        sys.stderr.write("reference_gpu: %r\n" % e)
        return None
    return {"value": round(r["mlups"], 1), "unit": "MLUPS",
            "kind": "the reference's HIP target (target_hip.c + its kernels), "
                    "this GPU, 10 steps after 1 warm-up",
            "ms_per_step": round(1e3 * r["t_total"] / r["steps"], 4)}


def dry_run(args, rank, world):
    """Everything of an N-rank run that needs no device: the ranks meet over
    gloo, cut the box, and every rank's X exchange schedule (lbmi_x_schedule,
    the list the library hands to RCCL) is checked against its neighbours':
    each send has its receive, same length, matched in order."""
    import torch.distributed as dist

    import ludwig_amd

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ntotal = tuple(args.size)
    if args.scaling == "weak":
        ntotal = (args.size[0] * world, args.size[1], args.size[2])
    dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, nhalo=args.nhalo)
    mine = ludwig_amd.x_schedule(args.nvel, dec.nlocal, args.nhalo, world, rank)
    every = [mine]
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, mine)
    ok = True
    for r in range(world):
        for peer in set(op["peer"] for op in every[r]):
            sends = [op["count"] for op in every[r]
                     if op["kind"] == "send" and op["peer"] == peer]
            recvs = [op["count"] for op in every[peer]
                     if op["kind"] == "recv" and op["peer"] == r]
            ok = ok and sends == recvs
    out = None
    if rank == 0:
        out = json.dumps({
            "dry_run": True, "n_gpus": world, "ranks_met": len(every),
            "schedules_pair_up": ok, "scaling": args.scaling,
            "config": {"workload": "D3Q%d %dx%dx%d, x-slab %d_1_1"
                       % (args.nvel, *ntotal, world),
                       "nlocal": list(dec.nlocal)},
            "bytes_per_exchange_per_rank": 8 * sum(
                op["count"] for op in mine if op["kind"] == "send")})
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("the X exchange schedules of the ranks do not pair up")
    return out


def peer_run(args):
    """--transport peer: the N-rank slab step in ONE process. Rank k is a host
    thread with a handle of its own on device k (modulo the devices present),
    the X planes travel as peer-to-peer copies between the devices (the peer
    ring of include/lbmi.h) instead of ncclSend / ncclRecv. Same step, same
    schedule, same three streams per rank as the RCCL path; the JSON line has
    the contract's fields, `transport: peer`, and per rank the phases of its
    step. A rank that fails or does not finish in time ends the run with a
    non-zero exit."""
    import threading

    import numpy as np
    import torch

    import ludwig_amd
    from ludwig_amd import synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no CPU path exists")
    world = args.gpus
    ndev = torch.cuda.device_count()
    ntotal = tuple(args.size)
    if args.scaling == "weak":
        ntotal = (args.size[0] * world, args.size[1], args.size[2])
    zeta = 0.3 if args.scheme == "m10" else 0.1
    lazy = (args.hydro == "lazy")
    ring = ludwig_amd.Ring(world)
    gate = threading.Barrier(world)
    res = [None] * world
    err = []
    m = ludwig_amd.lb.model(args.nvel)

    def rank_main(rank):
        try:
            dev = rank % ndev
            torch.cuda.set_device(dev)
            dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, nhalo=args.nhalo)
            lb = ludwig_amd.LB(args.nvel, dec.nlocal, args.nhalo, mode=ludwig_amd.FUSED,
                               halo_scheme=ludwig_amd.HALO_REDUCED, device=dev,
                               cartsz=world, cartrank=rank, own_stream=True)
            lb.relaxation_set(args.scheme, 0.1, zeta)
            for kv in filter(None, args.tune.split(",")):
                k, v = kv.split("=")
                lb.tune(k, int(v))
            lb.comm_init_ring(ring)
            synthetic.fill_device(lb, m["cv"], m["wv"], ntotal,
                                  xrange=(dec.noffset[0], dec.noffset[0] + dec.nlocal[0]))
            hydro = None
            if args.hydro != "0":
                hydro = ludwig_amd.Hydro(lb.nall, lb.device)
                hydro.force = torch.empty((3,) + lb.nall, dtype=torch.float64,
                                          device=lb.device)
                torch.cuda.synchronize(dev)
                lb.hydro_field_set(hydro.force, (0.0, 0.0, 0.0))
                if lazy:
                    lb.tune("hydro_lazy", 1)
                else:
                    lb.hydro_field_dirty(hydro.force)
            lb.synchronize()
            gate.wait()
            lb.run(hydro, PRELOAD_STEPS)
            lb.synchronize()
            mom0 = lb.moments()[[1, 5, 6, 7]]
            lb.run(hydro, 2 + args.warmup)
            lb.synchronize()
            lb.timing(max(1, args.steps // 6))
            gate.wait()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            lb.run(hydro, args.steps)
            if lazy:
                lb.hydro_sync()
            lb.synchronize()
            torch.cuda.synchronize(dev)
            gate.wait()
            dt = time.perf_counter() - t0
            kms, nlaunch = lb.timing_read()
            detail, ndetail = lb.timing_read_detail()
            lb.timing(False)
            mom1 = lb.moments()[[1, 5, 6, 7]]
            res[rank] = {"dt": dt, "mom0": mom0, "mom1": mom1, "nlocal": list(dec.nlocal),
                         "phases": {"rank": rank, "device": dev, "ring": world,
                                    "transport": "peer",
                                    "interior_ms": round(detail[0], 5),
                                    "exchange_ms": round(detail[1], 5),
                                    "boundary_ms": round(detail[2], 5),
                                    "steps_sampled": ndetail,
                                    "step_ms": round(kms / max(nlaunch, 1), 5)}}
            gate.wait()                  # nobody frees while a peer still reads
            lb.free()
        except Exception as e:           # noqa: BLE001
            err.append((rank, repr(e)))
            ring.abort()
            gate.abort()

    threads = [threading.Thread(target=rank_main, args=(r,), daemon=True)
               for r in range(world)]
    for t in threads:
        t.start()
    limit = time.time() + 900
    for t in threads:
        t.join(timeout=max(1.0, limit - time.time()))
    if err or any(t.is_alive() for t in threads) or any(r is None for r in res):
        sys.stderr.write("bench.py --transport peer: %r, %d rank(s) unfinished\n"
                         % (err, sum(t.is_alive() for t in threads)))
        sys.stderr.flush()
        os._exit(1)                      # (daemon threads may sit in a wait)
    ring.free()
    dt = max(r["dt"] for r in res)
    sites = ntotal[0] * ntotal[1] * ntotal[2]
    mom0 = sum(r["mom0"] for r in res)
    mom1 = sum(r["mom1"] for r in res)
    return json.dumps({
        "metric": "MLUPS (million lattice updates/sec) D3Q%d %dx%dx%d"
                  % (args.nvel, *ntotal),
        "value": round(1e-6 * sites * args.steps / dt, 1), "unit": "MLUPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "untimed_startup_steps": PRELOAD_STEPS + 2,
        "ms_per_step": round(1e3 * dt / args.steps, 5),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "D3Q%d %s single-fluid %dx%dx%d periodic, "
                        "lb_collide+lb_halo+lb_propagation per step"
                        % (args.nvel, args.scheme.upper(), *ntotal),
            "mode": "fused", "decomposition": "x-slab %d_1_1" % world,
            "transport": "peer: one process, one host thread per rank, "
                         "hipMemcpyPeerAsync between %d device(s)" % min(ndev, world),
            "halo": "reduced X planes, peer-to-peer copies",
        },
        "roofline": None, "cpu_baseline": None, "rccl_ranks": 0,
        "devices": min(ndev, world),
        "slab_step": [r["phases"] for r in res],
        "check": {"mass_drift_rel": float(abs(mom1[0] - mom0[0]) / mom0[0]),
                  "momentum_drift_abs": float(np.max(np.abs(mom1[1:] - mom0[1:])))},
    })


def kernel_source_sha1():
    import hashlib
    h = hashlib.sha1()
    for name in ("lbmi_kernels.hip", "lbmi_kernels.h"):
        with open(os.path.join(ROOT, "ludwig_amd", "csrc", name), "rb") as fp:
            h.update(fp.read())
    return h.hexdigest()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.transport == "peer" and args.gpus > 1 and world == 1:
        # one process drives all ranks: nothing to launch
        return peer_run(args)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            return self_launch(args)
        args.gpus = world
    if args.dry_run:
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist

    import ludwig_amd
    from ludwig_amd import synthetic

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no CPU path exists")
    torch.cuda.set_device(local_rank)

    if world > 1:
        # control plane only (barrier, max-reduction, id broadcast); the
        # data path is the library's own RCCL communicator (loopback
        # defaults for both are set at the top of this file)
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    ntotal = tuple(args.size)
    if args.scaling == "weak":
        ntotal = (args.size[0] * world, args.size[1], args.size[2])
    dec = ludwig_amd.SlabDecomposition(ntotal, world, rank, nhalo=args.nhalo)
    mode = {"fused": ludwig_amd.FUSED, "eager": ludwig_amd.EAGER,
            "inplace": ludwig_amd.INPLACE,
            "fused_soa": ludwig_amd.FUSED_SOA,
            "fused_halo": ludwig_amd.FUSED_HALO}[args.mode]
    if args.cartdim != 0 and not (world == 1 and args.selfring):
        raise SystemExit("--cartdim 1 | 2: with --selfring 1 on one GPU (the "
                         "N-rank runs of this bench cut along X)")
    lb = ludwig_amd.LB(args.nvel, dec.nlocal, args.nhalo, mode=mode,
                       halo_scheme=ludwig_amd.HALO_REDUCED, device=local_rank,
                       cartsz=world, cartrank=rank, cartdim=args.cartdim,
                       own_stream=bool(args.own_stream))
    zeta = 0.3 if args.scheme == "m10" else 0.1
    lb.relaxation_set(args.scheme, 0.1, zeta)
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        lb.tune(k, int(v))

    if world > 1:
        ids = [ludwig_amd.LB.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        lb.comm_init(ids[0])
    elif args.selfring:
        lb.comm_init(ludwig_amd.LB.comm_unique_id())

    noise_state = None
    if args.noise > 0.0:
        g = torch.Generator(device=lb.device)
        g.manual_seed(4321 + rank)
        nsite = lb.nall[0] * lb.nall[1] * lb.nall[2]
        noise_state = torch.randint(1, 2**31 - 1, (4, nsite), dtype=torch.int32,
                                    device=lb.device, generator=g)
        torch.cuda.synchronize()
        lb.noise_set(noise_state, args.noise, True)

    m = ludwig_amd.lb.model(args.nvel)
    synthetic.fill_device(lb, m["cv"], m["wv"], ntotal,
                          xrange=(dec.noffset[0], dec.noffset[0] + dec.nlocal[0]))
    hydro = None
    lazy = (args.hydro == "lazy")
    if args.hydro != "0":
        hydro = ludwig_amd.Hydro(lb.nall, lb.device)
        hydro.force = torch.empty((3,) + lb.nall, dtype=torch.float64,
                                  device=lb.device)
        torch.cuda.synchronize()
        # hydro_f_zero (ludwig.c:537) through the C-ABI, as the binding does:
        # the library then knows the array holds zeros
        lb.hydro_field_set(hydro.force, (0.0, 0.0, 0.0))
        lb.synchronize()
        if lazy:
            lb.tune("hydro_lazy", 1)
        if args.force_field or not lazy:
            # read by every collision: a per-site force field, or --hydro 1
            # (the reference reads hydro->force whatever it holds)
            lb.hydro_field_dirty(hydro.force)
        if args.force_field:
            # stands in for the thermodynamic force of config 4
            h = args.nhalo
            for a in range(3):
                n = lb.nlocal[a]
                off = dec.noffset[a]
                c = 1e-5 * torch.cos(2.0 * np.pi * (torch.arange(
                    n, dtype=torch.float64, device=lb.device) + off) / ntotal[a])
                shape = [1, 1, 1]
                shape[a] = n
                hydro.force[a][h:-h, h:-h, h:-h] = c.reshape(shape)
        torch.cuda.synchronize()

    fe = None
    if args.fe == "symmetric":
        if args.nhalo < 2:
            raise SystemExit("--fe symmetric needs --nhalo 2")
        if hydro is None:
            raise SystemExit("--fe symmetric needs --hydro 1")
        g = torch.Generator(device=lb.device)
        g.manual_seed(1234)
        fe = {"a": -0.00625, "b": 0.00625, "kappa": 0.004, "mobility": 1.25,
              "phi": torch.zeros(lb.nall, dtype=torch.float64, device=lb.device),
              "phi2": torch.zeros(lb.nall, dtype=torch.float64, device=lb.device)}
        h = args.nhalo
        fe["phi"][h:-h, h:-h, h:-h] = 0.05 * (torch.rand(
            lb.nlocal, dtype=torch.float64, device=lb.device, generator=g) - 0.5)
        if args.fe_route == "grad":
            fe["grad"] = torch.zeros((3,) + lb.nall, dtype=torch.float64,
                                     device=lb.device)
            fe["delsq"] = torch.zeros(lb.nall, dtype=torch.float64,
                                      device=lb.device)
        lb.fe_scheme_set(args.fe_grad, args.fe_order)
        if args.fe_route == "step" and (world > 1 or args.selfring or args.fe_halos
                                        or args.mode != "fused"):
            args.fe_route = "phi"
        fe["periodic"] = (world == 1 and not args.selfring
                          and args.fe_route in ("phi", "step") and not args.fe_halos)
        if args.fe_route == "step":
            # u of the previous collision / of this one: two arrays, swapped
            fe["u2"] = torch.zeros_like(hydro.u)
            # the thermodynamic force is the only contribution and stays in
            # registers: the array is zeroed through the library (not read)
            lb.hydro_field_set(hydro.force, (0.0, 0.0, 0.0))
            # u is stored by every step (the next one advects with it); rho,
            # which nothing on the device reads, is formed on demand -- as the
            # binding runs a free-energy case unconfigured (hydro_lazy 2)
            if args.fe_rho == "lazy":
                lb.tune("hydro_lazy", 2)
        torch.cuda.synchronize()

    def fe_step():
        # ludwig.c:537-791 for free_energy symmetric (finite difference).
        # hydro_f_zero is absorbed (the force kernel overwrites: no other
        # contribution exists in this configuration) and hydro_u_zero is
        # redundant on an all-fluid lattice (lb_collide writes u everywhere).
        if fe["periodic"]:
            # one rank: the pass wraps the periodic box by index, so neither
            # field_halo(phi) nor hydro_u_halo is needed
            lb.symmetric_step_periodic(fe["a"], fe["b"], fe["kappa"],
                                       fe["mobility"], fe["phi"], hydro.u,
                                       hydro.force, fe["phi2"], accumulate=False)
            fe["phi"], fe["phi2"] = fe["phi2"], fe["phi"]
            return
        lb.field_halo_n(fe["phi"], 2)                           # field_halo
        lb.field_halo_n(hydro.u, 1)                             # hydro_u_halo
        # phi_force_calculation + phi_cahn_hilliard, one pass over phi
        if args.fe_route == "grad":
            lb.field_grad(fe["phi"], fe["grad"], fe["delsq"])
            lb.symmetric_step_grad(fe["a"], fe["b"], fe["kappa"],
                                   fe["mobility"], fe["phi"], fe["grad"],
                                   fe["delsq"], hydro.u, hydro.force,
                                   fe["phi2"], accumulate=False)
        else:
            lb.symmetric_step(fe["a"], fe["b"], fe["kappa"], fe["mobility"],
                              fe["phi"], hydro.u, hydro.force, fe["phi2"],
                              accumulate=False)
        fe["phi"], fe["phi2"] = fe["phi2"], fe["phi"]

    def one_step():
        if fe is not None and args.fe_route == "step":
            # ludwig.c:537-860 in one call: force, Cahn-Hilliard, lb_collide,
            # lb_halo, lb_propagation
            uprev = hydro.u
            hydro.u, fe["u2"] = fe["u2"], hydro.u
            lb.symmetric_lb_step(hydro, uprev, fe["a"], fe["b"], fe["kappa"],
                                 fe["mobility"], fe["phi"], fe["phi2"])
            fe["phi"], fe["phi2"] = fe["phi2"], fe["phi"]
            return
        if fe is not None:
            fe_step()
        lb.step(hydro)

    def allsum(v):
        if world == 1:
            return v
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    # Start-up, not warm-up: the first launch of every kernel variant loads
    # its code object (4.6 ms for the first step, 1.11 for the second, 1.04
    # from the third on; tools/warmup_profile.py) and RCCL sets its channels
    # up on the first exchange. Three steps of the synthetic state get that
    # out of the way whatever --warmup is; the drift check starts after them.
    for _ in range(PRELOAD_STEPS):
        one_step()
    lb.synchronize()

    mom0 = allsum(lb.moments()[[1, 5, 6, 7]])
    # (the moments flush the deferred state: two more steps bring the
    # handle back to the steady state of the loop)
    one_step()
    one_step()

    for _ in range(args.warmup):
        one_step()
    lb.synchronize()
    # HIP events around every TIMING_PERIOD-th launch of the step's kernel
    # (each record costs the stream a few microseconds between two kernels)
    lb.timing(args.timing_period)

    barrier()
    torch.cuda.synchronize()
    lb.synchronize()
    t0 = time.perf_counter()
    if fe is None:
        # the three calls of every step issued by the C loop of the library
        # (lbmi_lb_run), as ludwig.c issues them, not one by one from Python
        lb.run(hydro, args.steps)
        if lazy:
            # rho, u of the last collision, as a run that reports statistics
            # at its end would ask for them: once, inside the timed region
            lb.hydro_sync()
    else:
        for _ in range(args.steps):
            one_step()
        if args.fe_route == "step" and args.fe_rho == "lazy":
            lb.hydro_sync()              # rho of the last step, asked for once
    lb.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    kms, nlaunch = lb.timing_read()
    detail, ndetail = lb.timing_read_detail()
    lb.timing(False)
    comm = lb.comm_info()
    phases = None
    if comm[0] > 0:
        # per rank: the three parts of a slab step, each on its own stream
        mine = {"rank": rank, "ring": comm[0],
                "transport": {0: "none", 1: "rccl", 2: "in-process"}[comm[2]],
                "interior_ms": round(detail[0], 5), "exchange_ms": round(detail[1], 5),
                "boundary_ms": round(detail[2], 5), "steps_sampled": ndetail,
                "step_ms": round(kms / max(nlaunch, 1), 5)}
        phases = [mine]
        if world > 1:
            phases = [None] * world
            dist.all_gather_object(phases, mine)
    order = {0: "soa", 1: "blocked [site/256][p][site%256] (deferred state)",
             2: "slot-swapped (AA)"}[lb.state()[2]]
    mom1 = allsum(lb.moments()[[1, 5, 6, 7]])

    sites = ntotal[0] * ntotal[1] * ntotal[2]
    mlups = 1e-6 * sites * args.steps / dt
    local_sites = dec.nlocal[0] * dec.nlocal[1] * dec.nlocal[2]
    pop_bytes = 2 * args.nvel * 8

    # After the timed region: the same loop with the hydro arrays read and
    # written by EVERY collision (what the reference's lb_collide does,
    # collision.c:329-333, 563-596): +24 B/site read, +32 B/site written.
    every = None
    if lazy and world == 1 and fe is None and not args.selfring:
        lb.tune("hydro_lazy", 0)
        lb.hydro_field_dirty(hydro.force)
        nextra = max(10, min(40, args.steps))
        for _ in range(3):
            one_step()                   # past the flush of the moments
        lb.synchronize()
        lb.timing(1)
        t1 = time.perf_counter()
        lb.run(hydro, nextra)
        lb.synchronize()
        dt1 = time.perf_counter() - t1
        kms1, nl1 = lb.timing_read()
        lb.timing(False)
        every = {
            "value": round(1e-6 * sites * nextra / dt1, 1), "unit": "MLUPS",
            "steps": nextra, "ms_per_step": round(1e3 * dt1 / nextra, 5),
            "bytes_per_lup": pop_bytes + 56,
        }
        if nl1 > 0 and kms1 > 0.0:
            # (EAGER has three kernels per step and no single timed launch)
            every["avg_launch_ms"] = round(kms1 / nl1, 5)
            every["achieved_GBs"] = round(1e-9 * (pop_bytes + 56) * local_sites
                                          / (1e-3 * kms1 / nl1), 1)
            every["populations_only_GBs"] = round(1e-9 * pop_bytes * local_sites
                                                  / (1e-3 * kms1 / nl1), 1)

    if rank == 0:
        # Algorithmic bytes per lattice update: one read and
        # one write of every population, 2*Q*8 B = 304 (D3Q19) / 432 (D3Q27):
        # THE figure of roofline.achieved and roofline.frac, whatever else the
        # kernel moves. With --hydro 1 the force read and the rho, u stores
        # (+56 B) are real traffic too: reported beside it, never in frac.
        roofline = None
        if nlaunch > 0 and args.mode in ("fused", "inplace", "fused_soa", "fused_halo"):
            t_launch = 1e-3 * kms / nlaunch
            achieved = 1e-9 * pop_bytes * local_sites / t_launch
            roofline = {
                "bound": "hbm", "kernel": "k_propagate_collide",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "bytes_per_lup": pop_bytes,
                "lups_per_launch": local_sites,
                "avg_launch_ms": round(1e3 * t_launch, 5),
                "launches": nlaunch,
                "launches_sampled": "every %d-th of %d" % (args.timing_period,
                                                          args.steps),
            }
            if args.mode == "fused_halo" and world == 1 and not args.selfring \
                    and args.noise == 0.0 and "halo_fold=0" not in args.tune:
                roofline["kernel"] = "k_propagate_collide_halo"
            if fe is not None and args.fe_route == "step":
                # the one kernel of the binary-fluid step: beside the
                # distributions rho, u stored (32 B), u of the previous step
                # (24 B) and phi in and out (16 B)
                extra = 72 - (8 if args.fe_rho == "lazy" else 0)
                roofline["kernel"] = "k_symm_lb_step"
                roofline["achieved_with_hydro_io"] = round(
                    1e-9 * (pop_bytes + extra) * local_sites / t_launch, 1)
                roofline["bytes_per_lup_with_hydro_io"] = pop_bytes + extra
            elif args.hydro == "1":
                roofline["achieved_with_hydro_io"] = round(
                    1e-9 * (pop_bytes + 56) * local_sites / t_launch, 1)
                roofline["bytes_per_lup_with_hydro_io"] = pop_bytes + 56
            # HBM bytes per launch from the PMC counters cannot be collected
            # inside this process (they need their own rocprofv3 --pmc
            # passes): quote the committed measurement of this command
            # (tools/profile_bench.sh + tools/pmc_summary.py), and only while
            # the kernel source it was taken with is the one that runs now
            pmc = os.path.join(ROOT, "profiles", "r03_default_pmc_hbm_traffic.json")
            same = (args.nvel == 19 and tuple(args.size) == (256, 256, 256)
                    and lazy and world == 1 and args.mode == "fused"
                    and args.scheme == "m10" and args.fe == "none"
                    and not args.selfring and not args.tune)
            if same and os.path.exists(pmc):
                with open(pmc) as fp:
                    t = json.load(fp)["summary"]
                if t.get("kernel_source_sha1") == kernel_source_sha1():
                    roofline["traffic"] = int(t["traffic_bytes"])
                    roofline["traffic_unit"] = "bytes per launch"
                    roofline["traffic_source"] = (
                        "profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / "
                        "WRITE_SIZE passes of this command, FETCH_SIZE x2 "
                        "(gfx950); algorithmic = %d"
                        % (os.path.basename(pmc), t["algorithmic_bytes"]))
                else:
                    roofline["traffic_source"] = (
                        "profiles/%s was taken with another kernel source: "
                        "not quoted" % os.path.basename(pmc))
        out = {
            "metric": "MLUPS (million lattice updates/sec) D3Q%d %dx%dx%d"
                      % (args.nvel, *ntotal),
            "value": round(mlups, 1),
            "unit": "MLUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "untimed_startup_steps": PRELOAD_STEPS + 2,
            "ms_per_step": round(1e3 * dt / args.steps, 5),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                # (the hydro traffic is part of the workload's name: rounds 1
                # and 2 reported different steps under one name -- r01 stored
                # rho, u and read the force every step, r02 did not)
                "workload": "D3Q%d %s single-fluid %dx%dx%d periodic, "
                            "lb_collide+lb_halo+lb_propagation per step, %s"
                            % (args.nvel, args.scheme.upper(), *ntotal,
                               ("binary fluid: force + Cahn-Hilliard + LB step in "
                                "one call; the thermodynamic force stays in "
                                "registers, u stored every step, rho %s"
                                % ("on demand" if args.fe_rho == "lazy"
                                   else "stored every step"))
                               if (fe is not None and args.fe_route == "step") else
                               {"lazy": "hydro arrays as the binding runs them "
                                "unconfigured: zero force field not read, "
                                "rho,u on demand (hydro_every_step: the "
                                "reference-equivalent step)",
                                "1": "force read and rho,u stored every step "
                                "(the reference-equivalent step)",
                                "0": "no hydro arrays"}[args.hydro]),
                "mode": args.mode,
                "order": order,
                "hydro_io": ("force array zeroed through the library, neither "
                             "written nor read; u stored by every step; rho %s"
                             % ("formed on demand, once, inside the timed region"
                                if args.fe_rho == "lazy" else "stored by every step"))
                if (fe is not None and args.fe_route == "step") else
                {"lazy": "arrays present; force zeroed through the "
                 "library (not read); rho,u formed on demand, once, "
                 "inside the timed region",
                 "1": "force read and rho,u stored by every collision",
                 "0": "no hydro arrays"}[args.hydro],
                "free_energy": args.fe if args.fe == "none" else
                "%s (%d-point gradients, advection order %d, from %s%s)"
                % (args.fe, args.fe_grad, args.fe_order,
                   {"step": "phi inside the LB kernel: force, Cahn-Hilliard "
                    "update and collision of a site by one thread "
                    "(lbmi_symmetric_lb_step), u stored every step, rho %s"
                    % ("on demand" if args.fe_rho == "lazy" else "stored every step")
                    }.get(args.fe_route, args.fe_route),
                   ", periodic wrap by index instead of field halos"
                   if fe["periodic"] else ""),
                "decomposition": ("x-slab %d_1_1" % world if args.cartdim == 0 else
                                  "%s-slab (1-rank ring along %s)"
                                  % ("xyz"[args.cartdim], "XYZ"[args.cartdim]))
                                 + (" (1-rank RCCL ring)" if args.selfring else ""),
                "halo": "index wrap (1 GPU); reduced X planes over RCCL (N>1)",
            },
                        "roofline": roofline,
            "hydro_every_step": every,
            "rccl_ranks": comm[0] if comm[2] == 1 else 0,
            "slab_step": phases,
            "check": {
                "mass_drift_rel": float(abs(mom1[0] - mom0[0]) / mom0[0]),
                "momentum_drift_abs": float(np.max(np.abs(mom1[1:] - mom0[1:]))),
            },
        }
        if args.cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
            out["reference_gpu"] = reference_gpu(args)
        else:
            out["cpu_baseline"] = None
        result = json.dumps(out)

    lb.free()
    if world > 1:
        dist.destroy_process_group()
    return result if rank == 0 else None


if __name__ == "__main__":
    # The contract is ONE JSON line on stdout. Gloo ("[Gloo] Rank 0 is
    # connected ...") and RCCL (its version banner) write to fd 1 from
    # native code: park fd 1 on stderr for the whole run and give it back
    # only for the line itself.
    sys.stdout.flush()
    _saved = os.dup(1)
    os.dup2(2, 1)
    _line = None
    try:
        _line = main()
    finally:
        sys.stdout.flush()
        os.dup2(_saved, 1)
        os.close(_saved)
    if _line is not None:
        print(_line, flush=True)
