import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from readbouncer_amd import capi, synth
dev = torch.device("cuda:0")
which = sys.argv[1]
N, L = int(os.environ.get("RB_N", "200000")), int(sys.argv[2])
mock = {}
for i, key in enumerate(("mock_deplete", "mock_t1", "mock_t2", "mock_t3")):
    mock[key] = synth.build_device_filter(0, synth.WORKLOADS[key], fill_seed=11 + i, plant_seed=110 + i, n_segments=512)[0]
dep, tgt = {"targets3": ([], ["mock_t1", "mock_t2", "mock_t3"]), "dt": (["mock_t3"], ["mock_t1"]), "t2": ([], ["mock_t1", "mock_t2"])}[which]
deplete, target = [mock[k] for k in dep], [mock[k] for k in tgt]
seqs, offs, lens = synth.make_reads_device(5, N, L, None, dev)
mc = torch.zeros((N, len(deplete) + len(target)), dtype=torch.int16, device=dev)
eng = capi.Engine(0, deplete, target)
if len(sys.argv) > 3 and sys.argv[3] == "nomerge":
    eng.set_merge(0)
print(which, L, eng.plan(0, N, L), flush=True)
eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr())
torch.cuda.synchronize()
print("classify ok", int(mc.max()), flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "cal":
    print(eng.calibrate(100000, L), flush=True)
    torch.cuda.synchronize()
print("done", flush=True)
