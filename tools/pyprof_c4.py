import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--config', 'c4', '--steps', '9', '--warmup', '3', '--no-cpu-baseline', '--no-roofline']
sys.path.insert(0, '.')
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
