import cProfile, pstats, sys, os, runpy
sys.argv = ["sar_ati_dcpa_csa_gpu.py", "--out", "/tmp/two.npz"]
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
runpy.run_path(os.path.join(root, "examples", "sar_ati_dcpa_csa_gpu.py"), run_name="__main__")   # warm (plans, first-touch)
pr = cProfile.Profile(); pr.enable()
runpy.run_path(os.path.join(root, "examples", "sar_ati_dcpa_csa_gpu.py"), run_name="__main__")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
