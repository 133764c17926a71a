"""GPU probe: two independent half-batches of bench.py's configuration on two engines and two HIP streams restricted to
DISJOINT CU sets (hipExtStreamCreateWithCUMask) against one engine and against the round-4 split on unrestricted streams. Hypothesis: the weight
nodes of a step are latency-bound (~1 TB/s), so two chains on disjoint halves of the chip overlap them; cost = the weights are streamed twice.
CU-mask bit -> CU mapping is not documented for gfx950: two disjoint patterns are tried (low / high 128 bits; alternating nibbles = bits whose
index % 8 is 0..3 / 4..7, an XCD split if the driver deals mask bits round-robin over the 8 XCDs).
  python tools/cumask_probe.py 128 64 -> profiles/r06_experiments.txt"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench

dev = torch.device("cuda:0")
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
    return torch.cuda.ExternalStream(st.value, device=dev)


PATTERNS = {
    "low128|high128": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
    "nibbles(i%8<4 | >=4)": ([0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8),
}

model = bench.build_model_on_device(dev, torch.bfloat16, "mini")


def run(B, tag):
    desc, prompt = bench.synthetic_batch(B, 0, dev)
    kw = dict(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_new_tokens=bench.NEW_TOKENS, min_new_tokens=bench.NEW_TOKENS)
    model.generate(**kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wav = model.generate(**kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[cumask_probe] B={B} {tag}: {dt * 1e3:.1f} ms per generate() = {B * bench.AUDIO_S / dt:.1f} audio-s/s", flush=True)
    return wav


for B in [int(x) for x in (sys.argv[1:] or ["128"])]:
    model.decode_streams = 0
    ref = run(B, "one engine")
    model.decode_streams = 2
    model.decode_streams_min_sub = 16
    model.__dict__.pop("_split_streams", None)
    w = run(B, "two half-batches, unrestricted streams (round 4)")
    print(f"[cumask_probe]   waveforms equal the single-engine run: {bool(torch.equal(w, ref))}", flush=True)
    for name, (m0, m1) in PATTERNS.items():
        model.__dict__["_split_streams"] = [masked_stream(m0), masked_stream(m1)]
        w = run(B, f"two half-batches, CU masks {name}")
        print(f"[cumask_probe]   waveforms equal the single-engine run: {bool(torch.equal(w, ref))}", flush=True)
    model.__dict__.pop("_split_streams", None)
    model.decode_streams = 0
    model._engine = None
