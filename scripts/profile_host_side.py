"""Host side of the eager step (per-GPU batch 1: the step time IS the host time): synchronising calls reported by
torch.cuda.set_sync_debug_mode, then a cProfile of 30 steps sorted by internal time.
    python scripts/profile_host_side.py [sync|profile]"""
import cProfile
import io
import os
import pstats
import sys
import warnings

mode = sys.argv[1] if len(sys.argv) > 1 else 'sync'
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-extras', '--steps', '6' if mode == 'sync' else '30', '--warmup', '3', '--batch', '1']
os.environ['GANGEALING_SYNTHETIC'] = '1'
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402

if mode == 'sync':
    import collections
    import traceback
    seen = collections.Counter()

    def hook(message, category, filename, lineno, file=None, line=None):
        if 'synchroniz' in str(message).lower():
            stack = [f for f in traceback.extract_stack() if '/gangealing_amd/' in f.filename or f.filename.endswith('bench.py')]
            key = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in stack[-3:])
            seen[key] += 1

    warnings.showwarning = hook
    warnings.simplefilter('always')
    torch.cuda.set_sync_debug_mode('warn')
    try:
        bench.main()
    except SystemExit:
        pass
    torch.cuda.set_sync_debug_mode('default')
    print('synchronising calls by call site (9 steps + set-up):')
    for k, v in seen.most_common(30):
        print(f'{v:6d}  {k}')
else:
    pr = cProfile.Profile()
    pr.enable()
    try:
        bench.main()
    except SystemExit:
        pass
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
    print(s.getvalue()[:9000])
