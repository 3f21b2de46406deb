"""BASELINE configs[0] on the CPU: testData/testQueries.fasta vs a 64-bin IBF in classify mode -- the plumbing case.
The counting is done by the oracle here (no GPU in this suite); the host plumbing that is exercised is the product's:
TOML surface (--dump-config), FASTA ingest (--parse-stats), fragmenter / filter sizing (C ABI host functions).
The same case runs end to end on the GPU in tests/test_host_cli.py::test_config1_build_and_classify."""
import os
import subprocess

import numpy as np

from oracle import pyoracle as po
from readbouncer_amd import capi
from tests import helpers as H
from tests.test_host_cli import CLI, run_cli, synth_genome, write_config


def test_config1_plumbing_and_oracle_driver(tmp_path, refdata):
    queries = os.path.join(refdata, "testQueries.fasta")
    (qid, qseq), = H.read_fasta(queries)
    # ingest: one record of 1890 bases
    kv = dict(x.split("=") for x in run_cli("--parse-stats", queries).stdout.split())
    assert (int(kv["records"]), int(kv["bases"])) == (1, 1890) and len(qseq) == 1890
    # the 64-bin filter: seeded 6.3 Mbp genome with the query's first kilobase planted (no E. coli genome offline)
    genome = synth_genome(1, 6_300_000, plant=qseq[:1000], at=2_345_678)
    cleaned = capi.cut_out_nnns(genome)
    n_bins = len(cleaned) // 100000 + 1
    starts, ends = capi.fragment_bounds(len(cleaned), 100000, 13)
    assert n_bins == 63 and len(starts) == 64  # 63 predicted bins; the 64th fragment is the k-mer-less tail
    bits = capi.calculate_filter_size_bits(100000, 13, 3, 0.01, n_bins)
    assert bits == 1236269 * 64
    o = H.build_filter_like_reference([genome], k=13, fragment_length=100000)
    assert (o.n_bins, o.n_bits, o.bin_width) == (n_bins, bits, 1)
    # TOML surface of the run (chunk_length 360, max_chunks 1 as in the reference's config.toml)
    ref_fa = tmp_path / "ecoli_like.fasta"
    ref_fa.write_text(">chr\n" + genome[:1000] + "\n")  # only its existence matters for the config check
    cfg = tmp_path / "c1.toml"
    write_config(cfg, "classify", tmp_path / "out", kmer_size=13, fragment_size=100000, deplete_files=[ref_fa],
                 read_files=[queries], chunk_length=360, max_chunks=1)
    dump = run_cli("--config", str(cfg), "--dump-config").stdout
    assert "chunk_length       = 360" in dump and "max_chunks         = 1" in dump and "testQueries.fasta" in dump
    # classify mode through the oracle's chunk driver: the planted query is depleted on its first 360 bp chunk
    res = po.classify_read_chunks([o], [], qseq, 360, 1)
    assert res == dict(status=po.OK, too_short=False, classified=True, best_target=-1, chunks_used=1)
    m = o.count_matches(po.encode(qseq[:360]))
    assert m == 348  # every 13-mer of the planted prefix is in its bin
    assert po.check_unblock([o], [], po.encode(qseq[:360])) == (po.OK, 1)
    # the unplanted tail of the query (bases 1000..1890) is not
    assert po.classify_read_chunks([o], [], qseq[1100:], 360, 2)["classified"] is False
