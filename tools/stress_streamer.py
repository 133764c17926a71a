"""GPU probe (not a test): repeats the streaming generate() scenario of tests/test_generate_gpu.py::test_streamer_thread_protocol
N times with a watchdog that dumps every thread's stack if one iteration exceeds 40 s (one hang in ~4 suite runs was observed)."""
import faulthandler, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases as C
import parler_tts_amd as P

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, *_ = C.tiny_model(seed=2)
m = m.to("cuda")
g = torch.Generator().manual_seed(3)
desc = torch.randint(3, 128, (1, 7), generator=g).cuda()
prompt_ids = torch.randint(3, 128, (1, 4), generator=g).cuda()
kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=120, min_new_tokens=120)
full = m.generate(**kw)[0].cpu().numpy()
for it in range(n_iter):
    faulthandler.dump_traceback_later(40, exit=True)
    for inc in (True, False):
        st = P.ParlerTTSStreamer(m, device="cuda", play_steps=20, stride=8, incremental=inc, timeout=60)
        errs = []
        def target():
            try:
                m.generate(streamer=st, **kw)
            except BaseException as e:
                errs.append(e); st.on_finalized_audio(np.zeros(0, dtype=np.float32), stream_end=True)
        th = threading.Thread(target=target); th.start()
        chunks = [c for c in st]
        th.join()
        if errs:
            print("iteration", it, "generate() raised:", repr(errs[0])); raise errs[0]
        audio = np.concatenate(chunks)
        assert audio.shape == full.shape, (audio.shape, full.shape)
    faulthandler.cancel_dump_traceback_later()
    print("iter", it, "ok", flush=True)
print("stress ok")
