"""Time k_minimizer_fast alone under the ablation switches of its debug instantiation (HULK_K1_DEBUG bits: 1 no list
stores, 4 no set (every candidate new), 8 phase A only, 16 no hash, 32 staging only, 64 bookkeeping only; 128 = none of them,
the debug instantiation itself)."""
import os, sys, subprocess, json
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, hulk_amd
    from hulk_amd import synth
    n = int(os.environ.get("K1N", "1000000"))
    sk = hulk_amd.GpuSketcher(21, 9, 8, stream=torch.cuda.current_stream().cuda_stream)
    b, o = synth.reads_torch(0, n, 150)
    torch.cuda.synchronize()
    for _ in range(3): sk.bin_reads_device(b.data_ptr(), o.data_ptr(), n, 150, b.numel())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sk.set_profiling(True)
    e0.record()
    for _ in range(20): sk.bin_reads_device(b.data_ptr(), o.data_ptr(), n, 150, b.numel())
    e1.record(); torch.cuda.synchronize()
    nl, ms = sk.get_profile("k_minimizer_fast")
    print(f"dbg={os.environ.get('HULK_K1_DEBUG','0'):>3}  whole K1 {e0.elapsed_time(e1)/20*1000/(n/100000):8.1f} us, "
          f"k_minimizer_fast alone {ms/max(nl,1)*1000/(n/100000):8.1f} us per 100k reads (n={n})")
else:
    for d in (0, 128, 128 + 1, 128 + 4, 128 + 5, 128 + 8, 128 + 16, 128 + 32, 128 + 64):
        env = dict(os.environ, HULK_K1_DEBUG=str(d), HULK_LIB="exp")   # (the switch exists in the profiling build only)
        subprocess.run([sys.executable, __file__, "child"], env=env)
