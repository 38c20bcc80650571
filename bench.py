#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on its named configuration.

  metric   : aligned candidate pairs/sec (whole node), 15 kb PacBio reads @15% error

  --mode locate (default)
  workload : BASELINE.json configs[1] -- 100 k synthetic 15 kb reads @15 % error against a 5 Mb synthetic
             genome, seed-hash + banded align on 1 MI355X (the locator.cpp path with R = 0.30, the
             reference's MAXR, because 15 %-error reads do not align at locator's hard-coded 0.15;
             mask 111*11*11*1*1111, 50 probe offsets, reads >= 500 bases) -- SURVEY.md 8d.
  step     : one pass of the hot path over the batch: seed-index build of the genome + the ordered
             first-success locate of every read (probe -> candidate pairs -> banded DP), inputs (packed
             genome and packed reads) already resident in HBM, result rows returned to the host.
  value    : candidate pairs the reference's loop hands to seq_aligner::align (all ranks) / wall time.
  scaling  : weak -- every rank owns its own 100 k reads; with N > 1 each rank scans 1/N of the genome's
             positions and the seed-index entries are all-gathered over RCCL/xGMI before every rank builds
             its lookup structure; the align step has no cross-GPU dependency.
  After the timed region the line also carries `overlap_strong`: one pass of the all-vs-all form below on
  --overlap-reads reads (0 = skip), so that a run at N = 1, 2, 4, 8 shows the strong-scaling curve too.

  --mode overlap
  workload : BASELINE.json configs[3] shape -- all-vs-all overlap of --overlap-reads synthetic 15 kb reads @15 %
             at 20x coverage, read shards across the GPUs: every rank generates and packs ITS shard,
             the packed shards are all-gathered once (setup, reported as read_gather_s), and a step is: probe
             entries of the rank's queries -> RCCL all-gather of the entry buffers -> probe table -> the rank's
             shard of the targets scanned, sorted and walked.  Strong scaling: the read set is fixed.

  python bench.py --gpus N --steps K --warmup W          one process per GPU; with N > 1 and no WORLD_SIZE in the
  environment this process starts the N ranks itself (as child processes, before anything touches the GPU) and
  relays rank 0's line; under torch.distributed.run it is one of the ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "aligned candidate pairs/sec (whole node), 15 kb PacBio reads @15% error"
HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md)
# wave64 VALU instructions per second the chip can issue: 256 CU x 4 SIMD-32, one wave64 instruction per 2 cycles per
# SIMD at 2.4 GHz (MI355X_MICROARCH.md, 'Wave scheduling'); tools/ubench_ops.hip measures 2.5 cycles for the plain
# integer opcodes (profiles/r01_ubench_valu_opcodes.txt), reported beside it
VALU_PEAK = 256 * 4 * 2.4e9 / 2.0
VALU_PEAK_MEASURED = 256 * 4 * 2.4e9 / 2.5
MASK_PAT = "111*11*11*1*1111"
EXIT_LEG_LOST = 3            # the headline line was printed, the extra all-vs-all leg hung or lost a rank
EXIT_HUNG = 4                # --mode overlap: the run did not finish within --timeout


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3,
                    help="untimed steps first (one of the first three calls on a fresh process takes ~25 ms longer on the host side)")
    ap.add_argument("--mode", choices=["locate", "overlap"], default="locate")
    ap.add_argument("--reads", type=int, default=100_000, help="locate mode: reads per GPU")
    ap.add_argument("--read-len", type=int, default=15_000)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--R", type=float, default=0.30)
    ap.add_argument("--trials", type=int, default=50)
    ap.add_argument("--kernel", choices=["auto", "rowsweep", "bitvec"], default="auto")
    ap.add_argument("--cpu-sample", type=int, default=512, help="reads timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores available to this process")
    ap.add_argument("--exchange", action="store_true",
                    help="take the multi-GPU seed-index exchange path (scan slice -> RCCL all-gather -> build) even at N=1")
    ap.add_argument("--overlap-reads", type=int, default=200_000,
                    help="all-vs-all: reads of the WHOLE set (strong scaling); in locate mode the size of the extra leg, 0 = skip it")
    ap.add_argument("--overlap-timeout", type=int, default=420,
                    help="locate mode: seconds after which the extra all-vs-all leg is given up and the headline line printed without it")
    ap.add_argument("--timeout", type=int, default=3000,
                    help="--mode overlap: seconds after which a rank that has not finished gives up with exit code 4")
    ap.add_argument("--overlap-check", type=int, default=3,
                    help="all-vs-all: targets of rank 0's shard re-done by the CPU oracle after the timed region (0 = skip)")
    ap.add_argument("--coverage", type=float, default=20.0, help="all-vs-all: genome = reads x read_len / coverage")
    ap.add_argument("--overlap-trials", type=int, default=32)
    ap.add_argument("--targets-per-call", type=int, default=40_000)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend; gloo + PBA_BENCH_SHARE_GPU=1 rehearses the N > 1 paths with several ranks "
                         "on ONE GPU (a 1-GPU box; RCCL itself needs one GPU per rank)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: every rank joins a gloo process group, the ranks are summed, rank 0 prints the census "
                         "(checks the launcher and the rendezvous on a box without GPUs)")
    return ap.parse_args()


def socket_cores() -> int:
    """Physical cores of socket 0 of this host (/proc/cpuinfo), 0 if it cannot be told."""
    try:
        cores, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = int(ln.split(":")[1])
            elif ln.startswith("core id"):
                core = int(ln.split(":")[1])
            elif not ln.strip():
                if phys == 0 and core is not None:
                    cores.add(core)
                phys = core = None
        return len(cores)
    except Exception:
        return 0


def host_cores() -> int:
    """CPU cores this process may really use: the affinity mask capped by the cgroup CPU quota (a 1-GPU box
    exposes every host core in the mask but grants a 16-core share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("PBA_MAX_HOST_THREADS", "64")))


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` called plainly: start the N ranks as child processes -- this process has not imported
    torch or touched the GPU, and never will -- wait for them, relay rank 0's JSON line.  Non-zero if any rank fails."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, rcs = b"", [None] * n
    try:
        # rank 0's stdout is drained while everybody runs (a full pipe would stall it); a rank that dies takes the
        # others down (they would wait for it in the next collective until the RCCL timeout)
        import threading
        buf = []
        t = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        t.start()
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            # (a rank that left with EXIT_LEG_LOST has only lost the extra all-vs-all leg: the others get to their own
            # fences -- rank 0 prints the headline line first -- and are not taken down)
            if any(rc not in (None, 0, EXIT_LEG_LOST) for rc in rcs):
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        p.terminate()
                deadline = time.time() + 10
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        try:
                            rcs[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                        except subprocess.TimeoutExpired:
                            p.kill()
                            rcs[r] = p.wait()
                break
            time.sleep(0.05)
        t.join(timeout=10)
        out0 = buf[0] if buf else b""
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        # a rank that lost only the extra all-vs-all leg (EXIT_LEG_LOST) has printed the headline line first: it is relayed,
        # and the failure stays visible in the exit code
        if lines and all(rc in (0, EXIT_LEG_LOST) for rc in rcs):
            print(lines[-1])
            return EXIT_LEG_LOST
        return 1
    if not lines:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(lines[-1])
    return 0


# ------------------------------------------------------------------------------------------------ all-vs-all leg
def overlap_leg(a, ctx, rank, world, nthreads, steps, warmup):
    """Strong-scaling all-vs-all (module docstring).  Returns the dict rank 0 reports (None on other ranks)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from pacbioassembly_amd import ProbeTable, distributed as pd, engine as eng

    if os.environ.get("PBA_BENCH_TEST_FAIL_RANK") == str(rank):       # tests: what a rank dying in this leg does to the line
        raise RuntimeError("PBA_BENCH_TEST_FAIL_RANK: this rank fails in the all-vs-all leg")
    n, rl, trials = a.overlap_reads, a.read_len, a.overlap_trials
    L = int(n * rl / a.coverage)
    mask = eng.mask_from_pattern(MASK_PAT)
    r_lo, r_hi = pd.shard_range(n, rank, world)            # this rank's reads: its queries AND its targets
    t0 = time.perf_counter()
    g = eng.synth_genome(2, L)
    text, offs = eng.synth_reads_range(3, g, r_lo, r_hi, rl, nthreads=nthreads)     # only the shard is generated ...
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    S = ctx.seqs_from_text(text, offs, strict_acgt=True)                            # ... and packed (on the GPU)
    del text, g
    t_pack = time.perf_counter() - t0
    t0 = time.perf_counter()
    if world > 1:                                          # all-gather of the packed shards, once per read set
        nb = S.packed_bytes
        buf = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
        my_offs = S.export(buf.data_ptr(), nb)
        allp, all_offs, all_lens = pd.all_gather_packed(buf[:nb], my_offs, S.lengths())
        S.close()
        S = ctx.seqs_from_device_packed(allp.data_ptr(), allp.numel(), all_offs, all_lens)
        del allp, buf
        torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0
    assert S.count == n
    cap = ((n + world - 1) // world) * 2 * trials + 64     # probe slots of the largest shard

    last = {}

    def step(t_hi=None):
        t_a = time.perf_counter()
        mine = torch.empty(cap, dtype=torch.int64, device="cuda")
        n_mine = ctx.overlap_probes(S, r_lo, r_hi, mask, trials, mine.data_ptr(), cap)
        if world > 1:
            probes, _ = pd.all_gather_entries(mine, n_mine)          # RCCL all-gather over xGMI: the one exchange
        else:
            mine[n_mine:] = pd.PAD
            probes = mine
        torch.cuda.synchronize()
        t_x = time.perf_counter() - t_a
        table = ProbeTable(ctx, probes.data_ptr(), probes.numel(), mask, trials)
        ov, st = ctx.overlap_all_sharded(S, mask, a.R, trials, 64, targets_per_call=a.targets_per_call, cap_per_target=400,
                                         t_lo=r_lo, t_hi=r_hi if t_hi is None else t_hi, table=table)
        table.close()
        st["exchange_s"] = t_x
        st["step_s"] = time.perf_counter() - t_a
        last["ov"] = ov
        return st

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step(t_hi=min(r_hi, r_lo + 256))                       # a sliver of the targets first: kernels, RCCL
    for _ in range(max(warmup, 1)):
        step()                                             # warm-up proper: a whole pass also sizes the ctx's work buffers
                                                           # (tens of GB of hipMalloc: 0.1 s of a 0.8 s pass at 200 k reads)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        st = step()
    fence()
    elapsed = time.perf_counter() - t0
    mine_t = [elapsed, st["exchange_s"], st["scan_ms"] * 1e-3, st["sort_ms"] * 1e-3, st["walk_ms"] * 1e-3, st["table_ms"] * 1e-3,
              t_gen, t_pack, t_gather]
    counts = [st["n_pairs"], st["n_overlaps"], st["n_candidates"]]
    if world > 1:
        tt = torch.tensor(mine_t, dtype=torch.float64, device="cuda")
        allt = torch.empty(world * tt.numel(), dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(allt, tt)
        allt = allt.view(world, -1).cpu().numpy()
        agg = torch.tensor(counts, dtype=torch.int64, device="cuda")
        dist.all_reduce(agg)
        counts = [int(x) for x in agg.tolist()]
    else:
        allt = np.array([mine_t])
    # ---- outside the timed region: the oracle on sampled targets of this rank's shard, every read of the set a query --
    # capacity-sized slices, the scan's 32 rows, the cascade of windows and k_ovl_after are all live at this size
    spot = None
    if rank == 0 and a.overlap_check > 0:
        spot = oracle_spot_check(ctx, S, last["ov"], r_lo, r_hi, n, rl, mask, a.R, trials, a.overlap_check, nthreads)
    S.close()
    if rank != 0:
        return None
    elapsed = float(allt[:, 0].max())
    visited = n * (min(rl - 16, 20000) + max(0, min(rl - 20016, 20000)))
    scan_s = float(allt[:, 2].max())
    # (tools/bench_overlap.py: 0.25 B of packed bases + 8 B of bucket offsets per visited position, 16 B of probe record per
    # candidate; the listed candidates' 8 B each are below 1 % of it)
    scan_bytes = visited * 8.25 + counts[2] * 16
    return {
        "workload": f"all-vs-all overlap (BASELINE configs[3] shape), {n} x {rl} reads @15%, {a.coverage}x coverage, R={a.R}, "
                    f"{trials} probe offsets per end, targets sharded over {world} GPU(s)",
        "scaling": "strong", "n_gpus": world, "world_size": dist.get_world_size() if world > 1 else 1, "steps": steps,
        "seconds_per_step": round(elapsed / steps, 4), "pairs_per_step": counts[0], "overlaps_per_step": counts[1],
        "candidates_per_step": counts[2], "pairs_per_s": round(counts[0] * steps / elapsed, 1),
        "overlaps_per_s": round(counts[1] * steps / elapsed, 1),
        "oracle_sample_identical": None if spot is None else spot["identical"], "oracle_sample": spot,
        "exchange_s": round(float(allt[:, 1].max()), 4), "read_gather_s": round(float(allt[:, 8].max()), 3),
        "per_rank": {"scan_s": [round(float(x), 4) for x in allt[:, 2]], "sort_s": [round(float(x), 4) for x in allt[:, 3]],
                     "walk_s": [round(float(x), 4) for x in allt[:, 4]], "table_s": [round(float(x), 4) for x in allt[:, 5]],
                     "generate_s": [round(float(x), 2) for x in allt[:, 6]], "pack_s": [round(float(x), 2) for x in allt[:, 7]]},
        "roofline_scan": {"bound": "hbm", "kernel": "k_ovl_scan", "unit": "GB/s", "peak": HBM_PEAK / 1e9,
                          "achieved": round(scan_bytes / world / scan_s / 1e9, 1) if scan_s > 0 else None,
                          "frac": round(scan_bytes / world / scan_s / HBM_PEAK, 5) if scan_s > 0 else None,
                          "note": "per GPU: 8.25 B per visited position + 16 B per candidate (its probe record), over the slowest rank's scan "
                                  "time; every candidate's first 32 rows run in the same kernel, only survivors are written"},
    }


def oracle_spot_check(ctx, S, ov, t_lo, t_hi, n, rl, mask, R, trials, k, nthreads):
    """The CPU oracle's locked round (orc_spaced_round: spaced_seed.cpp:420-437 with the intended seed_at) with target t as the
    reference and EVERY read of the set as a query, for k targets spread over [t_lo, t_hi): its successful rows must be the rows
    the GPU reported for t.  The read file of the oracle is put together from the packed reads on the device (uniform read
    length: [u32 length][packed bases] records, dna_seq.h:113-127)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oraclelib import Oracle
    t0 = time.perf_counter()
    nb = S.packed_bytes
    buf = torch.empty(max(nb, 1), dtype=torch.uint8, device="cuda")
    offs = S.export(buf.data_ptr(), nb)
    host = buf.cpu().numpy()
    del buf
    pk = (rl + 3) // 4
    stride = int(offs[1] - offs[0]) if n > 1 else pk
    assert (S.lengths() == rl).all() and stride >= pk and int(offs[0]) == 0
    rec = np.empty((n, 4 + pk), np.uint8)
    rec[:, :4] = np.frombuffer(np.uint32(rl).tobytes(), np.uint8)
    rec[:, 4:] = host[: n * stride].reshape(n, stride)[:, :pk]
    file = rec.tobytes()
    del rec, host
    rec_offs = np.arange(n, dtype=np.uint64) * np.uint64(4 + pk)
    orc = Oracle()
    picks = sorted({t_lo + (i * (t_hi - t_lo - 1)) // max(k - 1, 1) for i in range(k)})
    same, rows_checked = True, 0
    for t in picks:
        rows = orc.spaced_round(S.get_text(t), mask, R, file, rec_offs, trials, 64, buggy=False, nthreads=nthreads)
        exp = [(t, int(q), int(rows["j"][q]), int(rows["dir"][q]), int(rows["ref_pos"][q]), int(rows["cost"][q]), int(rows["matlen_a"][q]),
                int(rows["matlen_b"][q])) for q in np.nonzero(rows["found"])[0] if q != t]
        lo, hi = np.searchsorted(ov["target"], [t, t + 1])
        got = [tuple(int(x) for x in r) for r in ov[lo:hi]]
        same = same and got == exp
        rows_checked += len(exp)
    return {"identical": bool(same), "targets": picks, "overlap_rows": rows_checked, "queries_per_target": n - 1,
            "seconds": round(time.perf_counter() - t0, 1), "cores": nthreads}


def guarded_overlap(a, ctx, rank, world, nthreads, headline):
    """The all-vs-all leg of a default run, fenced off from the headline line: whatever happens in it -- an exception on
    this rank, or a collective that never returns because another rank failed -- rank 0 still prints `headline` (with the
    error noted in `overlap_strong`) FIRST, and the process then leaves with EXIT_LEG_LOST, never 0: a hang or a lost rank
    is visible to the launcher and in the run records.  (A single-rank exception is an ordinary result: the error is noted in
    the line and the run ends normally.)  headline: rank 0's finished line, None elsewhere."""
    import threading
    if a.overlap_reads <= 0:
        return None
    done = threading.Event()

    def leave(msg):
        if headline is not None:
            headline["overlap_strong"] = {"error": msg}
            print(json.dumps(headline), flush=True)
        print(f"bench.py rank {rank}: {msg}", file=sys.stderr, flush=True)
        os._exit(EXIT_LEG_LOST)                       # no teardown of a process group that may be wedged; never restarted

    def watchdog():
        if not done.is_set():
            leave(f"the all-vs-all leg did not finish within {a.overlap_timeout} s on rank {rank}")

    timer = threading.Timer(a.overlap_timeout, watchdog)
    timer.daemon = True
    timer.start()
    try:
        ov = overlap_leg(a, ctx, rank, world, nthreads, 1, 1)
    except BaseException as e:                        # noqa: BLE001 -- the headline line must not depend on this leg
        if world > 1:
            leave(f"rank {rank}: {type(e).__name__}: {e}")       # the other ranks are left to their watchdogs
        ov = {"error": f"{type(e).__name__}: {e}"}
    done.set()
    timer.cancel()
    return ov


# ------------------------------------------------------------------------------------------------ one rank
def bv_window(md: int, nb: int):
    """First-pass window of the bit-vector array (csrc/align_bitvec.h: bv_pass1_w / bv_pass1_wl)."""
    w = min(md, max(md // 2, md * 9 // 16) + 1)
    w = min(md, max(w, (2016 * nb + 64 - 4) * 2 // 3))
    return w, min(md, w // 2 + 1)


def run_rank(a):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist
    from pacbioassembly_amd import Context, build as pba_build, engine as eng
    from pacbioassembly_amd.engine import PBA_INDEX_ALL

    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py rank {rank}: needs an MI355X, there is no CPU path to time (no GPU visible)")
    if os.environ.get("PBA_BENCH_SHARE_GPU") == "1":                  # rehearsal: the ranks share the GPUs there are
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.exchange
    if use_dist:
        # RCCL writes its log to STDOUT (version banner, warnings about the host's kernel command line, ...), which must carry
        # ONE JSON line: the log goes to a file per process and is relayed to stderr when the process ends
        os.environ["NCCL_DEBUG"] = os.environ.get("PBA_NCCL_DEBUG", "WARN")
        rccl_log = os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/pba_rccl_{os.getpid()}.log")

        def relay_rccl_log():
            try:
                with open(rccl_log) as f:
                    sys.stderr.write(f.read())
                os.remove(rccl_log)
            except OSError:
                pass
        import atexit
        atexit.register(relay_rccl_log)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = Context(local_rank)
    nthreads = a.cpu_threads or max(1, host_cores() // (world if "LOCAL_WORLD_SIZE" in os.environ else 1))

    if a.mode == "overlap":
        # bounded like the default leg: a wedged collective (a peer gone) or a hung kernel ends the rank with EXIT_HUNG
        # after --timeout seconds instead of waiting for ever; an exception is an ordinary non-zero exit
        import threading

        def hung():
            print(f"bench.py rank {rank}: --mode overlap did not finish within {a.timeout} s", file=sys.stderr, flush=True)
            os._exit(EXIT_HUNG)
        timer = threading.Timer(a.timeout, hung)
        timer.daemon = True
        timer.start()
        ov = overlap_leg(a, ctx, rank, world, nthreads, a.steps, a.warmup)
        timer.cancel()
        if rank == 0:
            out = {"metric": METRIC, "value": ov["pairs_per_s"], "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
                   "warmup": a.warmup, "ms_per_step": round(1e3 * ov["seconds_per_step"], 3), "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                   "config": {"workload": ov["workload"], "reads": a.overlap_reads, "read_len": a.read_len, "coverage": a.coverage,
                              "R": a.R, "trials": a.overlap_trials, "mask": MASK_PAT,
                              "parallelism": f"read shards over {world} GPU(s), packed reads and probe entries all-gathered over RCCL"},
                   "overlap": ov}
            print(json.dumps(out))
        if use_dist:
            dist.destroy_process_group()
        return

    kernel = {"auto": eng.PBA_KERNEL_AUTO, "rowsweep": eng.PBA_KERNEL_ROWSWEEP, "bitvec": eng.PBA_KERNEL_BITVEC}[a.kernel]
    mask = eng.mask_from_pattern(MASK_PAT)

    # ---- synthetic inputs: genome seed 2, reads seed 3 (+ rank), 5/5/5 % ins/del/sub
    t0 = time.time()
    genome = eng.synth_genome(2, a.genome)
    reads, offs, _ = eng.synth_reads(3 + 1000 * rank, genome, a.reads, a.read_len, 0.05, 0.05, 0.05, nthreads=nthreads)
    t_gen = time.time() - t0
    cpu_reads = reads[: a.cpu_sample * a.read_len].copy() if rank == 0 else None
    t0 = time.time()
    T = ctx.seqs_from_text(genome, np.array([0, genome.size], np.uint64), strict_acgt=True)
    Rd = ctx.seqs_from_text(reads, offs, strict_acgt=True)        # H2D + 2-bit pack on the GPU: now resident
    t_up = time.time() - t0
    del reads

    def exchange_index():
        """N > 1: scan 1/N of the genome's positions, all-gather the entries over RCCL, build the lookup."""
        from pacbioassembly_amd import distributed as pd
        cap = pd.slice_capacity(a.genome, world)
        mine = torch.empty(cap, dtype=torch.int64, device="cuda")
        n_mine = ctx.index_scan(T, 0, mask, PBA_INDEX_ALL, rank, world, mine.data_ptr(), cap)
        allent, _ = pd.all_gather_entries(mine, n_mine)
        torch.cuda.synchronize()
        return ctx.index_from_entries(allent.data_ptr(), allent.numel(), mask, PBA_INDEX_ALL, a.genome)

    def step():
        ix = exchange_index() if use_dist else ctx.index_build(T, 0, mask, PBA_INDEX_ALL)
        prof_ix = ctx.last_profile()["index_ms"]
        rows, st = ctx.locate(ix, T, 0, Rd, a.R, a.trials, 500, kernel=kernel)
        prof = ctx.last_profile()
        prof["index_ms"] = prof_ix
        ix.close()
        return rows, st, prof

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    profs = []
    step_wall = []
    for _ in range(a.steps):
        t_s = time.perf_counter()
        rows, st, prof = step()
        profs.append(prof)
        step_wall.append(time.perf_counter() - t_s)
    fence()
    elapsed = time.perf_counter() - t0
    if os.environ.get("PBA_BENCH_VERBOSE"):
        print(f"rank {rank}: step wall ms {[round(1e3 * x, 2) for x in step_wall]}, kernel ms "
              f"{[round(p['align_ms'], 2) for p in profs]}", file=sys.stderr)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        agg = torch.tensor([st["n_pairs"], st["n_located"], st["n_cells"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(agg)
        pairs, located, cells = (int(x) for x in agg.tolist())
    else:
        pairs, located, cells = st["n_pairs"], st["n_located"], st["n_cells"]

    # ---- the all-vs-all leg (strong scaling) runs after the timed region of the headline metric, and after rank 0 has
    # put the headline line together (guarded_overlap below): nothing it does can cost the run its headline line
    Rd.close()
    T.close()
    if rank != 0:
        guarded_overlap(a, ctx, rank, world, nthreads, None)
        dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / a.steps
    value = pairs * a.steps / elapsed

    # ---- roofline of the dominant kernel (k_locate, first launch), per launch, this rank
    # algorithmic bytes (score-only align): per candidate pair the packed bases of both windows
    # plus a 24 B descriptor and a 28 B result; a = read from j (<= read_len), b = contig clipped to
    # len_a + max_dst (seq_aligner.h:94-102)
    md = 1 + int(a.read_len * a.R)
    bytes_per_pair = (a.read_len + 3) // 4 + (a.read_len + md + 3) // 4 + 24 + 28
    align_ms = float(np.mean([p["align_ms"] for p in profs]))
    redo_ms = float(np.mean([p["align_redo_ms"] for p in profs]))
    index_ms = float(np.mean([p["index_ms"] for p in profs]))
    algo_bytes = st["n_pairs"] * bytes_per_pair
    achieved = algo_bytes / (align_ms * 1e-3) / 1e9 if align_ms > 0 else 0.0
    # per-launch PMC figures from separate --pmc runs (tools/profile_bench.sh); they describe the build whose source
    # digest they carry and are dropped when the kernels have changed since
    traffic, pmc, pmc_note = None, {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            pmc = json.load(open(tpath))
            if pmc.get("source_digest") != pba_build.source_digest():
                pmc_note = (f"profiles/traffic.json was collected on source digest {pmc.get('source_digest')}, this build is "
                            f"{pba_build.source_digest()}: PMC-derived fields dropped (re-run tools/profile_bench.sh)")
                pmc = {}
            traffic = pmc.get("k_locate_hbm_bytes_per_launch")
        except Exception:
            traffic, pmc = None, {}
    nb1 = profs[-1]["nb_first"]
    roofline = {"bound": "hbm", "kernel": f"k_locate<{nb1}>", "achieved": round(achieved, 3),
                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 6),
                "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": round(align_ms, 3),
                "note": "score-only banded DP keeps its state in registers: HBM is not what bounds it (SURVEY 8d "
                        "expects <<1 %); the binding resource is integer VALU issue, see roofline_valu"}
    gcups = st["n_cells"] / (align_ms * 1e-3) / 1e9 if align_ms > 0 else 0.0
    # cells the array really processes for the located reads: m columns x the first-pass window (false candidates die
    # within their first 32-64 rows and are left out)
    win_cells = 0
    if nb1:
        w, wl = bv_window(md, nb1)
        sel = rows["found"] == 1
        win_cells = int((rows["seglen"][sel].astype(np.int64) * (w + wl + 1)).sum())
    valu_insts = pmc.get("valu_insts_per_launch")
    valu_rate = valu_insts / (align_ms * 1e-3) if valu_insts and align_ms > 0 else None
    roofline_valu = {"bound": "valu-issue", "achieved": round(valu_rate / 1e9, 1) if valu_rate else None,
                     "peak": round(VALU_PEAK / 1e9, 1), "unit": "G wave-instr/s",
                     "frac": round(valu_rate / VALU_PEAK, 4) if valu_rate else None,
                     "peak_measured_plain_int_ops": round(VALU_PEAK_MEASURED / 1e9, 1),
                     "frac_of_measured_peak": round(valu_rate / VALU_PEAK_MEASURED, 4) if valu_rate else None,
                     "reference_band_gcups": round(gcups, 1),
                     "processed_window_gcups": round(win_cells / (align_ms * 1e-3) / 1e9, 1) if align_ms > 0 else None,
                     "note": pmc_note or "peak = one wave64 VALU instruction per 2 cycles per SIMD (MI355X_MICROARCH.md); the instruction count "
                             "is the committed PMC run of this same build (profiles/traffic.json, source digest checked); "
                             "reference_band_gcups counts the cells of the reference's 2*max_dst+1 band, processed_window_gcups the "
                             "cells inside the window the array really sweeps for the located reads"}

    # ---- CPU baseline: the faithful oracle on this box's host cores, bounded sample of the same workload
    cpu = None
    if a.cpu_sample > 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oraclelib import Oracle
        orc = Oracle()
        ns = min(a.cpu_sample, a.reads)
        coffs = np.arange(ns + 1, dtype=np.uint64) * np.uint64(a.read_len)
        orc.prefault(nthreads, a.read_len, a.R)      # DP matrices mapped and touched before the clock starts
        t0 = time.perf_counter()
        crow, cst = orc.locator(genome, mask, a.R, cpu_reads[: ns * a.read_len], coffs, a.trials, 500,
                                nthreads=nthreads)
        ct = time.perf_counter() - t0
        same = all((crow[c] == rows[c][:ns]).all() for c in ("found", "j", "pos", "cost", "seglen", "matlen_a",
                                                              "matlen_b", "n_pairs"))
        sc = socket_cores()
        per_core = cst["n_pairs"] / ct / nthreads
        cpu = {"value": round(cst["n_pairs"] / ct, 3), "unit": "pairs/s", "cores": nthreads, "kind": "port",
               "sample": f"first {ns} reads of the same workload (index build + locate), {ct:.1f} s wall, "
                         f"{cst['n_pairs']} pairs, {cst['n_located']} located, {cst['n_cells'] / ct / 1e9:.2f} GCUPS "
                         f"(SURVEY 8d names the first 2 000 reads: ~4 x this sample's time on this box's {nthreads}-thread share, so the "
                         f"bounded sample stays at {ns})",
               # pairs/s per core and per socket.  One thread per core of this process's share; the socket figure
               # is the per-core rate times the socket's physical cores (an extrapolation: this box grants a share, not a socket)
               "pairs_per_s_per_core": round(per_core, 3), "socket_cores": sc or None,
               "socket_pairs_per_s_extrapolated": round(per_core * sc, 1) if sc else None,
               "gpu_vs_socket_extrapolated": round(value / (per_core * sc), 1) if sc else None,
               "gpu_rows_identical_on_sample": bool(same)}
        orc.release()

    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 100k synthetic 15 kb reads @15% error vs 5 Mb genome, "
                               "seed-hash + banded align (locator.cpp path, R=0.30, 50 probe offsets)",
                   "reads_per_gpu": a.reads, "read_len": a.read_len, "genome": a.genome, "R": a.R,
                   "trials": a.trials, "mask": MASK_PAT, "kernel": a.kernel,
                   "parallelism": f"reads sharded over {world} GPU(s)" + (", seed index all-gathered over RCCL" if use_dist else "")},
        "pairs_per_step": pairs, "located_per_step": located, "successful_pairs_per_s": round(located * a.steps / elapsed, 2),
        "band_gcups": round(cells * a.steps / elapsed / 1e9, 1),
        "band_gcups_note": "cells of the reference's 2*max_dst+1 band for the pairs aligned (what the CPU loop evaluates), not the "
                           "cells the array sweeps: see roofline_valu.processed_window_gcups",
        "kernel_ms": {"index_build": round(index_ms, 3), "locate_first": round(align_ms, 3),
                      "locate_redo": round(redo_ms, 3), "n_redo_reads": profs[-1]["n_redo"]},
        "setup_s": {"generate": round(t_gen, 2), "upload_and_pack": round(t_up, 2)},
        "source_digest": pba_build.source_digest(),
        "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu, "overlap_strong": None,
    }
    out["overlap_strong"] = guarded_overlap(a, ctx, rank, world, nthreads, out)
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def dry_run(a):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    from pacbioassembly_amd import distributed as pd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank, int(os.environ.get("LOCAL_RANK", "-1"))], dtype=torch.int64)
    dist.all_reduce(t)
    lo, hi = pd.shard_range(a.overlap_reads, rank, world)
    spans = torch.zeros(2 * world, dtype=torch.int64)
    dist.all_gather_into_tensor(spans, torch.tensor([lo, hi], dtype=torch.int64))
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size": dist.get_world_size(), "rank_sum": int(t[0]),
                          "local_rank_sum": int(t[1]), "read_shards": spans.view(world, 2).tolist()}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    if a.dry_run:
        dry_run(a)
    else:
        run_rank(a)


if __name__ == "__main__":
    main()
