"""Where does the launching thread spend its time in the encoder backward (it runs on autograd's worker thread, which a process-wide cProfile does not see)?
Wraps EncoderEngine.backward / forward in a cProfile of their own and runs the bench command.   python tools/host_profile.py > gpurun_out/host_profile.txt"""
import cProfile, pstats, sys, runpy, io, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import amd_pkg
amd_pkg.load()
from ts_asr_whisper_amd import engine
profs = {"backward": cProfile.Profile(), "forward": cProfile.Profile()}
calls = {"backward": 0, "forward": 0}


def wrap(name, fn):
    def w(self, *a, **k):
        p = profs[name]
        calls[name] += 1
        p.enable()
        try:
            return fn(self, *a, **k)
        finally:
            p.disable()
    return w


engine.EncoderEngine.backward = wrap("backward", engine.EncoderEngine.backward)
engine.EncoderEngine._forward_impl = wrap("forward", engine.EncoderEngine._forward_impl)
sys.argv = ["bench.py", "--steps", "8", "--warmup", "2", "--no-extra", "--no-cpu-baseline", "--no-power", "--no-one-stream-ref", "--profile-steps", "1"]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
for name, p in profs.items():
    s = io.StringIO()
    st = pstats.Stats(p, stream=s)
    st.sort_stats("tottime").print_stats(28)
    print(f"==== {name}: {calls[name]} calls\n" + s.getvalue())
