"""CPU: the arithmetic bench.py puts into the `roofline` object of its JSON line (algorithmic bytes per decode step of SURVEY.md
§8(d), node count of the captured step, dependency floor), with the GPU parts (engine, HIP-event timing) replaced by stand-ins.
A typo here would break the contract line on the GPU box, where nothing can be fixed any more."""
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _Eng:
    def decode_steps(self, n):
        self.n = getattr(self, "n", 0) + n


def _model(which, dtype=torch.bfloat16, fp8=False):
    cfg = bench.model_config(which)
    m = types.SimpleNamespace(config=cfg, dtype=dtype, decoder_weights_fp8=fp8)
    return m


@pytest.mark.parametrize("which,bs,step_us,nodes", [("mini", 1, 557.0, 122), ("mini", 32, 1291.0, 171), ("mini", 8, 1050.0, 170), ("large", 1, 940.0, 182)])
def test_roofline_object_arithmetic(monkeypatch, which, bs, step_us, nodes):
    monkeypatch.setattr(bench, "LIVE_PMC", False)  # no rocprofv3 child on the CPU tier: the committed pass is quoted
    monkeypatch.setattr(bench, "_prefilled_engine", lambda model, b, device, **gen: _Eng())
    monkeypatch.setattr(bench, "_timed_replays", lambda eng, n: step_us * 1e-6)
    r = bench.measure_decode_roofline(_model(which), bs, torch.device("cpu"))
    json.dumps(r)  # serialisable
    d = bench.model_config(which).decoder
    H, L, F, V, K = d.hidden_size, d.num_hidden_layers, d.ffn_dim, d.vocab_size, d.num_codebooks
    w_step = L * (6 * H * H + 2 * H * F) + K * V * H
    assert w_step == (362_348_544 if which == "mini" else 1_005_944_832)  # 362.3 M / 1005.9 M weights streamed per step
    lc = bench.N_PROMPT + 1 + 230 + 200
    expect = w_step * 2 + bs * 2 * L * H * (lc + bench.N_DESC) * 2 + bs * (K * H * 2 + K * V * 4)
    assert r["bytes_per_launch"] == expect and r["context"] == lc
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] == pytest.approx(expect / (step_us * 1e-6) / 1e9, rel=1e-3)
    assert r["frac"] == pytest.approx(r["achieved"] / 8000.0, abs=1e-4)
    assert r["us_per_launch"] == pytest.approx(step_us, abs=0.06)
    lm = r["latency_model"]
    assert lm["nodes"] == nodes and lm["floor_us_per_launch"] == pytest.approx(nodes * bench.NODE_FLOOR_US, abs=0.06)
    assert 0 < lm["frac_of_node_floor"] < 1 and lm["us_per_node"] == pytest.approx(step_us / nodes, abs=0.01)
    if which == "mini" and bs in (1, 32):  # committed PMC pass of this configuration
        assert isinstance(r["traffic"], int) and 0.9 < r["traffic"] / r["bytes_per_launch"] < 1.5  # the committed pass sits at the timed context
    else:
        assert r["traffic"] is None


def test_roofline_counts_one_byte_weights_for_the_fp8_gemv_step(monkeypatch):
    monkeypatch.setattr(bench, "LIVE_PMC", False)
    monkeypatch.setattr(bench, "_prefilled_engine", lambda model, b, device, **gen: _Eng())
    monkeypatch.setattr(bench, "_timed_replays", lambda eng, n: 1e-3)
    r8 = bench.measure_decode_roofline(_model("large", fp8=True), 4, torch.device("cpu"))
    r16 = bench.measure_decode_roofline(_model("large"), 4, torch.device("cpu"))
    assert r16["bytes_per_launch"] - r8["bytes_per_launch"] == 1_005_944_832  # e4m3 weights halve the weight term only
    r8b = bench.measure_decode_roofline(_model("large", fp8=True), 12, torch.device("cpu"))  # batch > 8: e4m3 MFMA strips stream the same bytes
    r16b = bench.measure_decode_roofline(_model("large"), 12, torch.device("cpu"))
    assert r16b["bytes_per_launch"] - r8b["bytes_per_launch"] == 1_005_944_832  # one byte per weight at EVERY batch size


def test_step_graph_node_counts_follow_the_forward_structure():
    # single utterance, folded cross block: qkv_attn + combine/out_proj + [xfold_attn | LN2+Mx, softmax+Up] + LN3/fc1 + fc2 (+ heads, tail)
    assert bench.step_graph_nodes(1, 24, 1024, True) == 122 and bench.step_graph_nodes(1, 30, 1536, True) == 182  # fused cross node up to hidden 1024
    assert bench.step_graph_nodes(1, 24, 1024, True, "f32") == 170 - 24  # fp32 at H = 1024: fused self-attention node, two-node cross block
    assert bench.step_graph_nodes(1, 24, 2048, True) == 170 and bench.step_graph_nodes(1, 24, 1024, False) == 146  # un-folded: qkv_attn + xq_attn
    assert bench.step_graph_nodes(4, 30, 1536, False) == 212 and bench.step_graph_nodes(8, 24, 1024, False) == 170  # 2..8 utterances: xq_attn_kernel, 7 per layer
    assert bench.step_graph_nodes(8, 24, 2048, False) == 194
    assert bench.step_graph_nodes(32, 24, 1024, False) == 171 and bench.step_graph_nodes(32, 24, 512, False) is None  # 7 nodes per layer + heads prep / heads / tail
    assert bench.step_graph_nodes(64, 24, 1024, False) == 171 and bench.step_graph_nodes(128, 30, 1536, False) == 213  # round 5: the 16-row LayerNorm + projection nodes above 40 utterances
    assert bench.step_graph_nodes(2, 24, 1024, False) == 146 and bench.step_graph_nodes(3, 24, 1024, False) == 146  # round 5: fused self-attention node up to 3 utterances


def test_live_pmc_result_replaces_the_committed_pass_and_failures_fall_back(monkeypatch):
    monkeypatch.setattr(bench, "_prefilled_engine", lambda model, b, device, **gen: _Eng())
    monkeypatch.setattr(bench, "_timed_replays", lambda eng, n: 6e-4)
    monkeypatch.setattr(bench, "LIVE_PMC", True)
    asked = []
    monkeypatch.setattr(bench, "measure_traffic_live", lambda bs, context, **k: asked.append((bs, context)) or {
        "traffic_bytes_per_step": 7.7e8, "context": context - 8, "algorithmic_mb": 776.6, "source": "LIVE in this bench run: rocprofv3 --pmc ..."})
    r = bench.measure_decode_roofline(_model("mini"), 1, torch.device("cpu"))
    assert r["traffic"] == 770000000 and r["traffic_note"].startswith("LIVE in this bench run")
    assert asked == [(1, r["context"])]  # measured at the context the step is timed at
    r = bench.measure_decode_roofline(_model("mini"), 1, torch.device("cpu"), live_pmc=False)  # side objects skip the child passes
    assert asked == [(1, r["context"])] and not r["traffic_note"].startswith("LIVE")
    monkeypatch.setattr(bench, "measure_traffic_live", lambda bs, context, **k: None)  # rocprofv3 missing / child failed: the committed pass is quoted
    r = bench.measure_decode_roofline(_model("mini"), 1, torch.device("cpu"))
    assert isinstance(r["traffic"], int) and not r["traffic_note"].startswith("LIVE")


def test_watchdog_prints_the_line_collected_so_far_and_exits_cleanly(tmp_path):
    """bench.py arms a watchdog before its optional side measurements: if one hangs (a dead streamer thread, a stuck HIP call), the
    process prints the JSON line with what has been measured and exits 0 - the contract numbers are never lost to a side measurement."""
    import subprocess
    import time

    script = tmp_path / "w.py"
    script.write_text(f"import sys, time\nsys.path.insert(0, {ROOT!r})\nimport bench\n"
                      "out = {'metric': 'm', 'value': 1.5, 'roofline': {'frac': 0.1}, 'cpu_baseline': {'value': 0.1}}\n"
                      "disarm = bench.arm_watchdog(out, 1.0)\nout['bs32'] = {'value': 2.0}\ntime.sleep(30)\nprint('never')\n")
    t0 = time.time()
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and time.time() - t0 < 25 and "never" not in r.stdout
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["value"] == 1.5 and j["bs32"] == {"value": 2.0} and "watchdog" in j and j["cpu_baseline"] == {"value": 0.1}
    script.write_text(f"import sys, time\nsys.path.insert(0, {ROOT!r})\nimport bench\nout = {{'value': 1.0}}\ndisarm = bench.arm_watchdog(out, 1.0)\n"
                      "disarm()\ntime.sleep(2.0)\nprint('finished normally')\n")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("finished normally") and "watchdog" not in r.stdout


def test_main_prints_one_contract_line_with_everything_stubbed(monkeypatch, capsys):
    """bench.main() end to end on the CPU with the GPU parts replaced (model, measurements, device calls): guards the assembly of the
    contract line itself - key names, order of the side measurements, watchdog arming / disarming - against edits made without a GPU."""
    import types

    import __graft_entry__ as ge

    class FakeModel:
        def __init__(self):
            self.calls = []

        def generate(self, **kw):
            self.calls.append(kw["input_ids"].shape[0])
            return torch.zeros(kw["input_ids"].shape[0], bench.FRAMES * 512)

    fake = FakeModel()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(ge, "build", lambda: None)
    monkeypatch.setattr(bench, "build_model", lambda rank, world, device, dtype, which="mini": fake)
    monkeypatch.setattr(bench, "synthetic_batch", lambda bs, rank, device: (torch.zeros(bs, bench.N_DESC, dtype=torch.long), torch.zeros(bs, bench.N_PROMPT, dtype=torch.long)))
    monkeypatch.setattr(bench, "measure_decode_roofline", lambda model, bs, device, live_pmc=True: {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None})
    monkeypatch.setattr(bench, "measure_fp32_parity_mode", lambda device: {"value": 12.0, "unit": "audio-seconds/sec"})
    monkeypatch.setattr(bench, "measure_large", lambda device: {"bf16_bs1": {"value": 9.0}})
    monkeypatch.setattr(bench, "measure_ttft", lambda model, bs, device, reps=20: 9.0)
    monkeypatch.setattr(bench, "measure_ttfa", lambda model, device: {"ttfa_p50_ms": 37.0})
    monkeypatch.setattr(bench, "measure_ttft_breakdown", lambda model, bs, device, reps=20: {"t5_ms": 1.0, "prefill_ms": 1.5, "bs": bs})
    monkeypatch.setattr(bench, "measure_sampling_step", lambda model, bs, device: {"ratio_sampling_over_greedy": 1.04})
    monkeypatch.setattr(bench, "cpu_baseline", lambda: {"value": 0.15, "unit": "audio-seconds/sec", "cores": 4, "kind": "port", "sample": "stub"})
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["unit"] == "audio-seconds/sec" and j["dtype"] == "bf16" and "workload" in j["config"] and "model" not in j["config"]
    assert j["cpu_baseline"]["kind"] == "port" and j["gpu_over_cpu"] == round(j["value"] / 0.15, 1)
    assert j["ttft"] == {"t5_ms": 1.0, "prefill_ms": 1.5, "bs": 1} and j["bs32"]["ttft"]["bs"] == 32
    assert j["bs32"]["roofline"]["bound"] == "hbm" and j["streaming"] == {"ttfa_p50_ms": 37.0} and j["sampling"]["ratio_sampling_over_greedy"] == 1.04
    assert "watchdog" not in j
    assert j["bs128"]["roofline"] == {"achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None} and j["bs128"]["unit"] == "audio-seconds/sec"
    assert j["fp32_parity_mode"]["value"] == 12.0 and j["fp32_parity_mode"]["gpu_over_cpu"] == 80.0 and j["large"] == {"bf16_bs1": {"value": 9.0}}
    # 1 warm-up + 2 timed steps at bs = 1, then the bs = 32 and bs = 128 side measurements (warm-up + timed each)
    assert fake.calls == [1, 1, 1, 32, 32, 128, 128]


def test_parity_tests_model_copy_equals_bench_build_model():
    """tests/test_bench_config_parity_gpu.py builds bench.py's model once (fp32, CPU) and hands every test a cast copy instead of paying
    the random init three times on the GPU box: the copy must be tensor-for-tensor what bench.build_model(dtype) returns."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_bench_config_parity_gpu as T

    _, got, _ = T._bench_model(torch.bfloat16, dev=torch.device("cpu"))
    want = bench.build_model(0, 1, torch.device("cpu"), torch.bfloat16)
    a, b = got.state_dict(), want.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) and a[k].dtype == b[k].dtype for k in a)
    wa, wb = got.audio_encoder._weights, want.audio_encoder._weights
    assert wa.keys() == wb.keys() and all(torch.equal(wa[k], wb[k]) for k in wa)
    assert got.dtype == torch.bfloat16 and got._engine is None and got is not T._MASTER[0]
    _, again, _ = T._bench_model(torch.float32, dev=torch.device("cpu"))
    assert len(T._MASTER) == 1 and again.dtype == torch.float32 and torch.equal(again.state_dict()["embed_prompts.weight"], T._MASTER[0].state_dict()["embed_prompts.weight"])


@pytest.mark.parametrize("extra", [[], ["--model", "large", "--bs", "1"], ["--model", "large", "--dtype", "fp8w", "--bs", "4"]])
def test_self_launch_builds_the_drivers_command_line(extra, monkeypatch):
    """`python bench.py --gpus 8 [...]` must run unmodified when an 8-GPU node appears. Without WORLD_SIZE in the environment main()
    re-launches itself with exactly the command the driver uses (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>), one rank per GPU, dmabuf IPC on - checked here by capturing the command instead of running it."""
    import subprocess

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return 0

    argv = ["--gpus", "8", "--steps", "3", "--warmup", "1"] + extra
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", fake_call)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:4] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv  # the ranks see the caller's arguments, nothing added or dropped
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["PTTS_BENCH_LAUNCHED"] == "1"


def test_gpus_flag_must_agree_with_the_process_group(monkeypatch):
    """A launcher that starts 2 ranks while the command line says --gpus 4 is refused before any work (the line's n_gpus would otherwise lie)."""
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert "WORLD_SIZE=2" in str(ei.value.code)


_RANK_SCRIPT = r'''
import json, os, sys, types
sys.path.insert(0, {root!r})
import torch
import bench
import __graft_entry__ as ge

class FakeModel:
    def generate(self, **kw):
        return torch.zeros(kw["input_ids"].shape[0], bench.FRAMES * 512)

torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 2
torch.cuda.set_device = lambda i: None
torch.cuda.synchronize = lambda *a: None
ge.build = lambda: None
bench.build_model = lambda rank, world, device, dtype, which="mini": FakeModel()
bench.synthetic_batch = lambda bs, rank, device: (torch.zeros(bs, bench.N_DESC, dtype=torch.long), torch.zeros(bs, bench.N_PROMPT, dtype=torch.long))
bench.measure_ttft = lambda model, bs, device, reps=20: 2.0 + int(os.environ["RANK"])
bench.measure_decode_roofline = lambda model, bs, device, live_pmc=True: {{"bound": "hbm", "frac": 0.1, "live": live_pmc}}
sys.argv = ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--bs", "3"]
bench.main()
'''


def test_two_ranks_on_gloo_print_one_line_with_n_gpus_from_the_process_group(tmp_path):
    """bench.main() as two real processes under torch.distributed.run (gloo, CPU; model and device calls stubbed): exactly ONE JSON line (rank 0's),
    n_gpus = the process group's world size, value = the units of all ranks / the slowest rank's time, per-rank objects for both ranks, the
    time-to-first-token of the line = the slowest rank's p50, and no profiler child passes on a multi-rank run."""
    import socket
    import subprocess

    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PTTS_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["world_size_reported_by_process_group"] == 2 and j["config"]["global_batch"] == 6 and j["config"]["backend"] == "gloo"
    assert len(j["per_rank_ms_per_step"]) == 2 and len(j["per_rank_value"]) == 2
    assert abs(j["value"] - 2 * 3 * 2 * bench.AUDIO_S / (max(j["per_rank_ms_per_step"]) * 2 / 1e3)) / j["value"] < 2e-3
    assert j["per_rank_ttft_p50_ms"] == [2.0, 3.0] and j["ttft_p50_ms"] == 3.0 and j["roofline"]["live"] is False and j["scaling"] == "weak"
