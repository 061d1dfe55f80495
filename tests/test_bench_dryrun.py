"""bench.py's multi-rank plumbing without a GPU: `python bench.py --gpus 2 --dry-run` re-executes itself
under torch.distributed.run (127.0.0.1), initialises the process group (gloo here, nccl = RCCL on the GPU box), runs the lagged
token all-gather with its drain, the phase marks, the max-over-ranks reduction and prints ONE JSON line from rank 0 -- the code
path of the real run with the kernels stubbed out in the bench only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    # the driver's contract: stdout holds ONE line, it is the last one, it fits well inside the driver's 8 KB stdout tail (the
    # round-5 line had grown to 26 KB and BENCH_r05.json's `parsed` was null) and it is STRICT JSON (no NaN / Infinity tokens)
    # (gloo's own "[Gloo] Rank r is connected ..." chatter may precede it on stdout for world > 1)
    lines = p.stdout.splitlines()
    assert lines and lines[-1].startswith("{") and sum(ln.startswith("{") for ln in lines) == 1, p.stdout[-2000:]
    assert len(lines[-1].encode()) < 6144, len(lines[-1])

    def _no_const(tok):
        raise AssertionError(f"non-strict JSON token {tok}")
    return json.loads(lines[-1], parse_constant=_no_const)


def test_final_line_budget_on_a_full_size_record():
    """The compacting path on the LARGEST record the bench has produced (profiles/r05_bench_line.json: ViT-L + nested InternViT-6B,
    19 + 20 roofline entries, 26 KB): the final line stays under 6 KB, keeps what the driver and the review read (value,
    ms_per_step, config.workload, roofline, rooflines.{attn, msda, gemm...}.frac, cpu_baseline, internvit6b.*) and is strict JSON."""
    import contextlib
    import io
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    full["rooflines"]["attn"]["frac"] = float("nan")          # a poisoned figure must not produce a NaN token
    extra = full.pop("internvit6b")
    head = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    out, err = io.StringIO(), io.StringIO()
    detail = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    had = os.path.exists(detail)
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
        bench.emit(dict(full, internvit6b=extra), head, full, extra)
    if not had and os.path.exists(detail):
        os.remove(detail)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0].encode()) < 6144, [len(x) for x in lines]
    rec = json.loads(lines[0], parse_constant=lambda t: (_ for _ in ()).throw(AssertionError(t)))
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["config"]["workload"].startswith("vitl14")
    assert set(rec["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert rec["rooflines"]["attn"]["frac"] is None and rec["rooflines"]["msda"]["frac"] > 0
    assert set(rec["rooflines"]["gemm"]) == {"frac", "us_per_launch", "launches_per_step", "traffic"}
    assert "dcnv3" not in rec["rooflines"] and "dcnv3" in rec["isolated_frac"]         # isolated 8(f) rows: fraction only
    assert rec["cpu_baseline"]["cores"] >= 1 and len(rec["cpu_baseline"]["sample"]) <= 300
    iv = rec["internvit6b"]
    assert iv["value"] > 0 and iv["roofline"]["frac"] > 0 and iv["rooflines"]["attn"]["frac"] > 0
    assert err.getvalue().startswith("bench detail: {")


def test_bench_respawns_and_runs_two_ranks_on_cpu():
    rec = _run("--gpus", "2")
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    cfg = rec["config"]
    assert cfg["rccl_ranks"] == 2 and cfg["backend"] == "gloo" and cfg["parallelism"] == "dp2+allgather(tokens)"
    assert "DRY RUN" in cfg["workload"]                      # a plumbing line must never pass for a measurement
    # headline = the collective waited for inside its step; the pipelined schedule is reported next to it
    assert cfg["allgather_lag_steps"] == 0 and rec["allgather_lag_alt"]["allgather_lag_steps"] == 1
    assert rec["value"] > 0 and rec["allgather_lag_alt"]["value"] > 0
    assert set(rec["phases_ms"]) >= {"vit_projector", "token_allgather", "msda_12_calls"}
    assert "rooflines" not in rec and "cpu_baseline" not in rec


def test_bench_dry_run_single_rank_and_direct_allgather():
    rec = _run()
    assert rec["n_gpus"] == 1 and rec["config"]["rccl_ranks"] == 1 and "allgather_lag_alt" not in rec
    rec = _run("--gpus", "2", "--allgather", "direct", "--allgather-lag", "1")
    assert rec["config"]["allgather"] == "direct" and rec["config"]["allgather_lag_steps"] == 1
    assert rec["allgather_lag_alt"]["allgather_lag_steps"] == 0


def test_bench_dry_run_world_4_and_8_ragged_shards_both_algorithms_both_lags():
    """the multi-GPU path at the world sizes the driver's SCALE run uses, on gloo: ragged tile counts (1-7
    tiles per image dealt to the ranks by dist.shard_images), the collective waited for inside the step (lag 0) and one step
    later (lag 1), RCCL-style collective and direct all-peers point-to-point.  In a dry run every step's tokens carry (rank,
    step), and bench.py checks every gathered tensor slice by slice: `dry_run_collectives_checked` counts them."""
    for world, algo, lag in ((4, "collective", 0), (4, "direct", 1), (8, "collective", 1), (8, "direct", 0)):
        rec = _run("--gpus", str(world), "--ragged-tiles", "--allgather", algo, "--allgather-lag", str(lag))
        cfg = rec["config"]
        assert rec["n_gpus"] == world and cfg["rccl_ranks"] == world and cfg["allgather"] == algo and cfg["allgather_lag_steps"] == lag
        counts = cfg["tiles_per_rank"]
        assert len(counts) == world and len(set(counts)) > 1 and max(counts) - min(counts) <= 2   # ragged, balanced by tile count
        assert sum(counts) == sum(1 + (i * i + 2 * i + 1) % 7 for i in range(world * 8))
        # every rank reports its own clock and phases
        assert [r["rank"] for r in rec["per_rank"]] == list(range(world))
        assert [r["tiles"] for r in rec["per_rank"]] == counts
        assert all(r["timed_s"] > 0 for r in rec["per_rank"])
        # warm-up 1 + 2 timed steps x 2 schedules + 3 phase steps = 8 collectives, every one checked on rank 0
        assert rec["dry_run_collectives_checked"] == 8, rec["dry_run_collectives_checked"]
        assert rec["allgather_lag_alt"]["allgather_lag_steps"] == 1 - lag
