"""bench.py's one-line JSON contract, on the GPU box (small sizes)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout: %r" % lines[:5]
    return json.loads(lines[0])


def test_single_gpu_line():
    j = _run(["--steps", "3", "--warmup", "1", "--pairs", "3000000", "--cpu-sample", "20000", "--e2e-pairs", "30000", "--e2e-chunks", "1",
              "--strong-sample", "1000000", "--config-steps", "3"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert (j["n_gpus"], j["steps"], j["warmup"], j["higher_is_better"], j["scaling"], j["vs_baseline"]) == \
        (1, 3, 1, True, "weak", None)
    assert 50 <= j["untimed_launches"] <= 400 and j["untimed_launches"] % 25 == 0 and j["world"] == 1 and j["launched_by"] == "single process"
    assert len(j["ranks"]) == 1 and j["ranks"][0]["device"] == 0
    assert j["dtype"] == "u8" and j["data"] == "synthetic" and j["unit"] == "read-pairs/s"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["verified"] is True
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - 3000000 * 34 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["matches_gpu_codes"] is True
    assert abs(j["value"] - 3000000 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    # the three rates of SURVEY.md 8(d) and both CPU baselines travel in the same line
    assert r["traffic_source"] is None or "replayed" in r["traffic_source"]
    st = c["strong"]
    assert st["matches_gpu_codes"] is True and st["one_core"]["cores"] == 1 and st["one_core"]["value"] > 0
    assert st["all_cores"]["cores"] == st["cores_available"] >= 1 and st["cpu"]
    sm = j["extra"]["streamed"]
    assert sm["codes_ok"] is True and sm["value"] > 0 and sm["h2d_GBps"] > 0 and sm["batches"] >= 10
    e = j["extra"]["e2e"]
    assert e["pairs"] == 30000 and e["value"] > 0 and e["gzip_level"] == 1 and e["gzip_backend"] in ("libdeflate", "zlib")
    assert e["counts_total_pass_fail_undetermined"][0] == 30000 == sum(e["counts_total_pass_fail_undetermined"][1:])
    assert e["cpu_seconds"] > 0 and 0 < e["core_utilisation"] <= 1.05
    assert e["host_pool_only"]["value"] > 0 and e["host_pool_only"]["counts_equal"] is True and e["host_pool_only"]["gzip_level"] == 1
    h = e["huffman_only"]
    assert h["gzip_level"] == -1 and h["value"] > 0 and h["counts_equal"] is True and h["output_gz_bytes"] > 0
    assert h["host_pool_only"]["value"] > 0 and h["host_pool_only"]["counts_equal"] is True
    # the reference's real input format (one gzip member per file) and the driver's default output level, same line
    sm1 = e["single_member_gzip"]
    assert sm1["value"] > 0 and sm1["counts_equal"] is True and sm1["gzip_level"] == 1 and sm1["vs_bgzf_input"] > 0
    assert e["binned_qualities"]["value"] > 0 and e["binned_qualities"]["counts_equal"] is True
    assert e["host_level6"]["gzip_level"] == 6 and e["host_level6"]["counts_equal"] is True and e["default_level"] == 1
    # hygiene: what ran as warm-up, the step-based fraction beside the event-based one
    assert j["warmup_ran"] == j["untimed_launches"] and 0 < r["frac_by_step"] <= r["frac"] * 1.001
    assert "traffic_age_commit" in r
    # the single-member files went through the device's gzip kernels, none was handed to the host; the chunks were distinct
    assert e["distinct_chunks"] is True and "hot" in e["input_page_cache"]
    assert sm1["pipeline"]["gzip_members"] == 4 and sm1["pipeline"]["gzip_fallbacks"] == 0 and sm1["pipeline"]["text_segments"] == 0, sm1["pipeline"]
    # the kernel on the other BASELINE configs, verified, in the same line
    kc = j["extra"]["kernel_configs"]
    assert sorted(kc) == ["cfg2", "cfg4", "cfg5", "kit8u12x2", "kit8u9"]  # (the kits: on their static shapes of the fast kernel)
    assert kc["kit8u12x2"]["kernel"] == "demux_fast" and kc["kit8u9"]["kernel"] == "demux_fast"
    for name, k in kc.items():
        assert k["verified"] is True and k["kernel_ms"] > 0 and 0 < k["frac"] < 1, (name, k)
        assert abs(k["frac"] - k["pairs"] * k["algorithmic_bytes_per_pair"] / (k["kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-9


def test_two_ranks_self_spawned():
    """`python bench.py --gpus 2` with NO launcher (how the driver may start it): bench.py starts its
    own two ranks as child processes and relays one line with n_gpus 2.  Both ranks on GPU 0 and gloo
    for the count reduce here (one GPU on this box; RCCL refuses two ranks on one device)."""
    env = dict(os.environ, QUADE_BENCH_DEVICE="0", QUADE_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    j = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "2000000"], env=env)
    assert j["n_gpus"] == 2 and j["world"] == 2 and j["verified"] is True and j["launched_by"] == "self-spawn"
    assert [r["rank"] for r in j["ranks"]] == [0, 1] and all(r["kernel_ms"] > 0 for r in j["ranks"])
    assert j["count_reduce"]["backend"] == "gloo" and j["count_reduce"]["ms_max_over_ranks"] > 0
    assert abs(j["value"] - 2 * 2000000 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]


def test_two_rank_rehearsal_line():
    """torch.distributed.run with 2 ranks; both on GPU 0 and gloo for the count reduce (this box has
    one GPU) -- the driver's multi-GPU runs use one GPU per rank and the nccl (RCCL) backend."""
    env = dict(os.environ, QUADE_BENCH_DEVICE="0", QUADE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", "29577"]
    j = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "2000000", "--e2e-pairs", "40000", "--e2e-rank-chunks", "2"], env=env, launcher=launcher)
    assert j["n_gpus"] == 2 and j["verified"] is True and "cpu_baseline" not in j
    assert j["launched_by"] == "external launcher" and j["world"] == 2
    assert abs(j["value"] - 2 * 2000000 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    # the demultiplexing leg of the N > 1 line: chunk-sharded fastq.gz -> fastq.gz through the product's multi-rank path
    er = j["extra"]["e2e_ranks"]
    assert "error" not in er, er
    assert er["world"] == 2 and er["chunks"] == 4 and er["pairs"] == 160000 and er["value"] > 0 and er["distinct_chunks"] is True
    assert er["counts_total_equal"] is True and er["total_pairs_in_report"] == 160000
    assert [r["rank"] for r in er["ranks"]] == [0, 1] and [r["chunks"] for r in er["ranks"]] == [[0, 2], [1, 3]]
    assert all(r["pairs"] == 80000 and r["pipeline"]["host_inflated_runs"] == 0 for r in er["ranks"]), er["ranks"]
    assert er["rehearsal_on_one_gpu"] is True and "files" in er["count_reduce"]["backend"]  # (the driver's runs: "rccl via qd_reduce_counts")


def test_single_rank_rccl_path():
    """The calls of the N > 1 path (process group on the nccl = RCCL backend, barrier, int64 SUM and
    float64 MAX all-reduce) with one rank, and a clean stdout despite RCCL's banner."""
    env = dict(os.environ, QUADE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                "--master-addr", "127.0.0.1", "--master-port", "29578"]
    j = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--pairs", "2000000", "--no-cpu-baseline"], env=env,
             launcher=launcher)
    assert j["n_gpus"] == 1 and j["verified"] is True
