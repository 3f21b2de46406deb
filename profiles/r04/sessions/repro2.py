import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from readbouncer_amd import capi, synth
dev = torch.device("cuda:0")
N, L = 1_000_000, 250
mock = {}
for i, key in enumerate(("mock_deplete", "mock_t1", "mock_t2", "mock_t3")):
    mock[key] = synth.build_device_filter(0, synth.WORKLOADS[key], fill_seed=11 + i, plant_seed=110 + i, n_segments=512)[0]
seq = sys.argv[1].split(",")
mode = sys.argv[2] if len(sys.argv) > 2 else ""
for step, which in enumerate(seq):
    dep, tgt = {"readme": (["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"]), "targets3": ([], ["mock_t1", "mock_t2", "mock_t3"]),
                "dt": (["mock_t3"], ["mock_t1"]), "t2": ([], ["mock_t1", "mock_t2"]), "d1": (["mock_deplete"], [])}[which]
    deplete, target = [mock[k] for k in dep], [mock[k] for k in tgt]
    seqs, offs, lens = synth.make_reads_device(5, N, L, None, dev)
    mc = torch.zeros((N, len(deplete) + len(target)), dtype=torch.int16, device=dev)
    eng = capi.Engine(0, deplete, target)
    if mode == "nomerge" and step > 0:
        eng.set_merge(0)
    if mode == "nophase" and step > 0:
        eng.set_phased(0, 0, 0, 0, 1)
    if mode == "nooverlap":
        eng.set_overlap(False)
    for it in range(3):
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr())
        torch.cuda.synchronize()
    print(step, which, "ok", eng.merge_info(), int(mc.max()), flush=True)
    if mode != "nodestroy":
        eng.destroy()
    if mode == "keeptensors":
        keep = globals().setdefault("_keep", [])
        keep.append((seqs, offs, lens, mc))
print("done", flush=True)
