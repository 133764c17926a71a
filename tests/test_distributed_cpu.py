"""CPU, world_size 2 over gloo: the N>1 plumbing of the generation path — static utterance sharding, the one-off
bucketed weight broadcast (rank 0 -> all) and the max-over-ranks timing reduction bench.py uses. No data-path
collective exists to test: utterances are independent."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from parler_tts_amd.distributed import shard_batch, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from parler_tts_amd.distributed import broadcast_tensors, max_over_ranks, shard_batch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(123)
        ref = [torch.randn(257, 33, generator=g), torch.randn(1000, generator=g).to(torch.bfloat16), torch.arange(17, dtype=torch.int64),
               torch.randn(3, 5, 7, generator=g)]
        mine = [t.clone() if rank == 0 else torch.zeros_like(t) for t in ref]
        calls = broadcast_tensors(mine, src=0, bucket_bytes=20_000)  # forces several buckets + mixed dtypes per bucket
        ok = all(torch.equal(a, b) for a, b in zip(mine, ref))
        batch = torch.arange(9 * 4).reshape(9, 4)
        part, none = shard_batch([batch, None], rank, world)
        gathered = [torch.zeros(5, 4, dtype=batch.dtype) for _ in range(world)]
        pad = torch.zeros(5, 4, dtype=batch.dtype)
        pad[: part.shape[0]] = part
        dist.all_gather(gathered, pad)  # test-side check only; the product path never gathers
        t = max_over_ranks(1.0 + rank)
        q.put((rank, ok, calls, part.shape[0], none is None, [x.tolist() for x in gathered], t))
    finally:
        dist.destroy_process_group()


def test_world_size_2_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ok0, calls0, n0, none0, g0, t0), (r1, ok1, calls1, n1, none1, g1, t1) = res
    assert ok0 and ok1, "rank 1 did not receive rank 0's weights bit-exactly"
    assert calls0 == calls1 and calls0 >= 3
    assert (n0, n1) == (5, 4) and none0 and none1
    batch = torch.arange(9 * 4).reshape(9, 4)
    rebuilt = torch.tensor(g0[0])[:5].tolist() + torch.tensor(g0[1])[:4].tolist()
    assert rebuilt == batch.tolist()
    assert t0 == t1 == 2.0


def _model_worker(rank, world, port, q):
    """what bench.py does at N > 1 (build_model): every rank constructs the model from ITS OWN random state, rank 0's weights win"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    from cases import tiny_model
    from parler_tts_amd.distributed import broadcast_model_weights

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 draws its weights; rank 1 is built the way bench.py builds ranks != 0: allocated WITHOUT initialisation (text encoder
        # under transformers' no_init_weights, its embedding tie redone so the parameter lists match), then overwritten by the broadcast
        m, _, _, _ = tiny_model(seed=0, init_weights=(rank == 0))
        if rank != 0:  # a replica that must be overwritten: different decoder / text-encoder / prompt-embedding / codec tensors
            with torch.no_grad():
                for p in m.parameters():  # uninitialised memory may hold anything (NaN included): give it defined, wrong values
                    if p.is_floating_point():
                        p.nan_to_num_(0.0, 0.0, 0.0)
            torch.manual_seed(999)
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(torch.randn_like(p) if p.is_floating_point() else 1)
                m.audio_encoder._weights = {k: v + 1 for k, v in m.audio_encoder._weights.items()}
        m._engine, m.audio_encoder._engine = "stale", "stale"  # packed copies of the old tensors must be dropped
        n = broadcast_model_weights(m, src=0, bucket_bytes=1 << 20)
        ref, _, _, _ = tiny_model(seed=0)  # rank 0's tensors, rebuilt locally from the same seed
        same = all(torch.equal(a, b) for (_, a), (_, b) in zip(sorted(m.state_dict().items()), sorted(ref.state_dict().items())))
        same_dac = all(torch.equal(m.audio_encoder._weights[k], v) for k, v in ref.audio_encoder._weights.items())
        q.put((rank, n, same, same_dac, m._engine is None and m.audio_encoder._engine is None, len(list(m.parameters()))))
    finally:
        dist.destroy_process_group()


def test_world_size_2_model_weight_broadcast():
    """`broadcast_model_weights` on a whole ParlerTTSForConditionalGeneration (T5 + decoder holder + prompt embedding + DAC wrapper's
    tensor dict): rank 1 ends with rank 0's tensors bit for bit, in few collectives, and the lazily packed engines are invalidated."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, same0, dac0, inv0, np0), (_, n1, same1, dac1, inv1, np1) = res
    assert same0 and dac0, "rank 0's own weights changed"
    assert same1 and dac1, "rank 1 did not receive rank 0's weights bit-exactly"
    assert inv0 and inv1
    assert n0 == n1 and n0 < np0, f"{n0} collectives for {np0} parameter tensors: small tensors must be coalesced"


def _sharded_generate_worker(rank, world, port, q):
    """`distributed.generate_sharded` over gloo: every rank holds the same oracle-backed stand-in model (tests/test_generate_glue_cpu.py)
    and the FULL batch; rank 0 must get what the single-process call returns, in input order, with the per-utterance lengths."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist

    from test_generate_glue_cpu import _model
    from parler_tts_amd.distributed import generate_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        m, _, _, _ = _model(eos_gain=6.0)  # EOS reachable: utterances of different lengths -> ragged blocks, zero padding
        g = torch.Generator().manual_seed(1)
        desc, prompt_ids = torch.randint(3, 128, (3, 9), generator=g), torch.randint(3, 128, (3, 4), generator=g)
        mask = torch.ones(3, 9, dtype=torch.long)
        mask[1, 6:] = 0
        kw = dict(input_ids=desc, attention_mask=mask, prompt_input_ids=prompt_ids, do_sample=False, max_length=40, min_new_tokens=3)
        res = {}
        for dst in (0, None):
            out = generate_sharded(m, dst=dst, **kw)
            res[dst] = None if out is None else (out.sequences.clone(), list(out["audios_length"]))
        grp = dist.new_group([0, 1])
        sub = generate_sharded(m, dst=1, group=grp, **kw)  # an explicit group, result on ITS rank 1
        one = generate_sharded(m, dst=0, input_ids=desc[:1], prompt_input_ids=prompt_ids[:1], do_sample=False, max_length=24, min_new_tokens=3)  # rank 1 idles
        # num_return_sequences carried by a `generation_config=` OBJECT (not a kwarg): every rank returns twice the rows of its slice; the block size of
        # the gather comes from what generate() returned (sized from the arguments, one rank overflowed its metadata and the others hung)
        import copy

        gc2 = copy.deepcopy(m.generation_config)
        gc2.num_return_sequences, gc2.do_sample, gc2.max_length, gc2.min_new_tokens = 2, True, 24, 3  # (transformers rejects greedy + num_return_sequences > 1)
        kw2 = dict(input_ids=desc, attention_mask=mask, prompt_input_ids=prompt_ids, generation_config=gc2)
        nrs = generate_sharded(m, dst=0, **kw2)
        single = m.generate(return_dict_in_generate=True, **kw) if rank == 0 else None
        single1 = m.generate(return_dict_in_generate=True, input_ids=desc[:1], prompt_input_ids=prompt_ids[:1], do_sample=False, max_length=24,
                             min_new_tokens=3) if rank == 0 else None
        ok = True
        if rank == 0:
            for dst in (0, None):
                w, lens = res[dst]
                ok = ok and lens == list(single["audios_length"]) and w.shape == single.sequences.shape and torch.equal(w, single.sequences.float().cpu())
            ok = ok and one is not None and list(one["audios_length"]) == list(single1["audios_length"]) and torch.equal(one.sequences, single1.sequences.float().cpu())
            ok = ok and len(set(single["audios_length"])) > 1  # the case really is ragged
            ok = ok and sub is None
            ok = ok and nrs is not None and nrs.sequences.shape[0] == 6 and len(nrs["audios_length"]) == 6  # (sampled: shards draw their own streams)
        else:
            ok = res[0] is None and res[None] is not None and one is None
            ok = ok and sub is not None and list(sub["audios_length"]) == res[None][1] and torch.equal(sub.sequences, res[None][0])
        q.put((rank, bool(ok), None if res[None] is None else res[None][1]))
    finally:
        dist.destroy_process_group()


def test_world_size_2_generate_sharded_equals_the_single_process_call():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_generate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1], "rank 0: the gathered waveforms / lengths differ from the single-process generate()"
    assert res[1][1], "rank 1: wrong return protocol (None for dst=0, the full result for dst=None)"
    assert res[0][2] == res[1][2] and len(res[0][2]) == 3  # dst=None: every rank holds the same lengths, in input order
