"""Host side (C++ over the C ABI): TOML surface of ReadBouncer's [IBF] section, FASTA/FASTQ reader, and the
build / classify usages of the CLI.  CPU tests cover parsing and error behaviour; the GPU tests run
BASELINE config 1 (testData/testQueries.fasta vs a 64-bin IBF) end to end and compare with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "readbouncer_amd", "readbouncer_amd_cli")


def run_cli(*args, check=True):
    p = subprocess.run([CLI] + list(args), capture_output=True, text=True)
    if check and p.returncode != 0:
        raise AssertionError("cli failed rc=%d\nstdout:%s\nstderr:%s" % (p.returncode, p.stdout, p.stderr))
    return p


def write_config(path, usage, out_dir, **ibf):
    lines = ['usage = "%s"' % usage, "output_directory = '%s'" % out_dir, "log_directory = '%s/logs'" % out_dir, "", "[IBF]"]
    for k, v in ibf.items():
        if isinstance(v, (list, tuple)):
            lines.append("%s = [%s]" % (k, ", ".join("'%s'" % x for x in v)))
        else:
            lines.append("%s = %s" % (k, v))
    path.write_text("\n".join(lines) + "\n")


def test_cli_exists():
    assert os.path.exists(CLI), "run __graft_entry__.build()"


# Same keys, sections and values as the config.toml at the root of the reference repository (the file its
# libConfigReaderTests parse), written out here with our own layout: inline comments, mixed quote styles,
# integer and float values, string arrays and an integer array, all three tables.
REFERENCE_STYLE_CONFIG = """
usage = "test"   # one of build / target / classify / test
output_directory = '{out}'
log_directory = '{out}/logs'

[IBF]
kmer_size = 15            # default 13
fragment_size = 100000
threads = 3
target_files = ['{t0}', '{t1}', '{t2}']
deplete_files = ['{d0}']
read_files = ['{r0}']
exp_seq_error_rate = 0.1  # between 0 and 1
chunk_length = 360
max_chunks = 1

[MinKNOW]
host = "localhost"
port = "9502"
flowcell = "MS00000"
token_path = "test/tmp/minknow-auth-token.json"
channels = [1,512]

[Basecaller]
caller = "DeepNano"
host = "127.0.0.1"
port = "9502"
threads = 3
config = "dna_r9.4.1_450bps_fast"
"""


def test_reference_style_config_is_parsed(tmp_path):
    names = dict(t0="path/to/reference/file/Bacillus_subtilis_complete_genome.fasta", t1="b.fasta", t2="c.fasta",
                 d0="d.fasta", r0="r.fasta")
    cfg = tmp_path / "config.toml"
    cfg.write_text(REFERENCE_STYLE_CONFIG.format(out=tmp_path / "RB_out", **names))
    # listed genome paths that do not exist -> the reference's error text (configReader.cpp:292-297)
    p = run_cli("--config", str(cfg), "--dump-config", check=False)
    assert p.returncode == 1
    assert "[Error] The following target file does not exist: path/to/reference/file/Bacillus_subtilis_complete_genome.fasta" in p.stderr
    assert (tmp_path / "RB_out" / "logs").is_dir()  # parse_general creates both directories
    # same file with existing inputs: every value comes through
    for k in names:
        f = tmp_path / (k + ".fasta")
        f.write_text(">x\nACGTACGTACGTACGTACGT\n")
        names[k] = str(f)
    cfg.write_text(REFERENCE_STYLE_CONFIG.format(out=tmp_path / "o2", **names))
    out = run_cli("--config", str(cfg), "--dump-config").stdout
    assert 'usage              = "test"' in out
    assert "kmer_size          = 15" in out and "fragment_size      = 100000" in out and "threads            = 3" in out
    assert "chunk_length       = 360" in out and "max_chunks         = 1" in out
    assert "exp_seq_error_rate = 0.10000000000000001" in out
    assert out.count(".fasta'") == 5


def test_defaults_and_errors(tmp_path):
    ref = tmp_path / "ref.fasta"
    ref.write_text(">r\n" + "ACGT" * 50 + "\n")
    cfg = tmp_path / "c.toml"
    write_config(cfg, "build", tmp_path / "out", target_files=[ref])
    out = run_cli("--config", str(cfg), "--dump-config").stdout
    # parser defaults, configReader.cpp:238-243
    for line in ("kmer_size          = 13", "fragment_size      = 100000", "threads            = 1",
                 "chunk_length       = 250", "max_chunks         = 5", "exp_seq_error_rate = 0.10000000000000001"):
        assert line in out
    # no filter files at all
    write_config(cfg, "classify", tmp_path / "out")
    p = run_cli("--config", str(cfg), "--dump-config", check=False)
    assert p.returncode == 1 and "At least one target or deplete file has to be specified" in p.stderr
    # classify needs read_files
    write_config(cfg, "classify", tmp_path / "out", deplete_files=[ref])
    p = run_cli("--config", str(cfg), "--dump-config", check=False)
    assert p.returncode == 1 and "read_files" in p.stderr
    # missing top-level key
    cfg.write_text('usage = "build"\noutput_directory = "%s"\n' % (tmp_path / "out"))
    p = run_cli("--config", str(cfg), check=False)
    assert p.returncode == 1 and "log_directory" in p.stderr
    # usage "target" without chunk files would need the live MinKNOW connection: refused with a pointer to the replay
    write_config(cfg, "target", tmp_path / "out", deplete_files=[ref])
    p = run_cli("--config", str(cfg), check=False)
    assert p.returncode == 2 and "outside this engine's scope" in p.stderr and "read_files" in p.stderr
    # unsupported usage is refused, not silently ignored
    write_config(cfg, "test", tmp_path / "out", deplete_files=[ref])
    p = run_cli("--config", str(cfg), check=False)
    assert p.returncode == 2 and "outside this engine's scope" in p.stderr
    p = run_cli(check=False)
    assert p.returncode == 1


def _fnv(recs):
    h = 1469598103934665603
    for rid, seq in recs:
        for ch in (rid + "\t" + seq + "\n").encode():
            h = ((h ^ ch) * 1099511628211) & (2**64 - 1)
    return h


def test_ingest_parser_matches_python_reader(tmp_path, refdata):
    cases = [(os.path.join(refdata, "classifyTests_test.fastq"), H.read_fastq),   # CRLF FASTQ
             (os.path.join(refdata, "libIBFTests_test1.fasta"), H.read_fasta),    # multi-line FASTA, blank line
             (os.path.join(refdata, "testQueries.fasta"), H.read_fasta)]
    weird = tmp_path / "w.fasta"
    weird.write_bytes(b"\n\n>a desc\r\nACGT\r\nAC\r\n\r\nGT\n>b\n>c\nNNNN\n>d\nAC")  # empty record, no final newline
    cases.append((str(weird), H.read_fasta))
    mixed = tmp_path / "m.fq"
    mixed.write_bytes(b"@r1 x\nACGTN\n+\n!!!!!\n@r2\nAC\n+r2\n@@\n")  # quality line starting with '@'
    cases.append((str(mixed), H.read_fastq))
    # a FASTQ whose quality lines love '@' and '>' and '+': every line start is a tempting record boundary
    rng = np.random.default_rng(8)
    tricky = tmp_path / "tricky.fq"
    with open(tricky, "wb") as fh:
        for i in range(400):
            L = int(rng.integers(1, 90))
            seq = H.random_dna(rng, L, with_n=0.05).encode()
            qual = bytes(rng.choice(np.frombuffer(b"@>+I!", dtype=np.uint8), size=L))
            fh.write(b"@r%d extra\n%s\n+%s\n%s\n" % (i, seq, b"" if i % 3 else b"r%d" % i, qual))
    cases.append((str(tricky), H.read_fastq))
    long_fa = tmp_path / "long.fa"
    with open(long_fa, "w") as fh:
        for i in range(120):
            seq = H.random_dna(rng, int(rng.integers(0, 400)))
            fh.write(">s%d\n" % i + "\n".join(seq[j:j + 60] for j in range(0, len(seq), 60)) + "\n")
    cases.append((str(long_fa), H.read_fasta))
    for path, reader in cases:
        recs = reader(path)
        # serial-equivalent (one segment) and cut into many small segments parsed by several threads: same records in
        # the same order
        for extra in ((), ("--ingest-threads", "3", "--segment-bytes", "64"), ("--ingest-threads", "2", "--segment-bytes", "1000"),
                      ("--ingest-threads", "5", "--segment-bytes", "4096")):
            out = run_cli(*extra, "--parse-stats", path).stdout.split()
            kv = dict(x.split("=") for x in out)
            assert int(kv["records"]) == len(recs), (path, extra)
            assert int(kv["bases"]) == sum(len(s) for _, s in recs), (path, extra)
            assert int(kv["fnv"]) == _fnv(recs), (path, extra)
    kv = dict(x.split("=") for x in run_cli("--ingest-threads", "3", "--segment-bytes", "4096", "--parse-stats", str(tricky)).stdout.split())
    assert int(kv["segments"]) > 3  # the file really was cut
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r1\nACGT\nIIII\n")
    assert run_cli("--parse-stats", str(bad), check=False).returncode == 1
    bad2 = tmp_path / "bad2.fq"  # malformed record deep inside a multi-segment file: reported, not skipped
    bad2.write_bytes(open(tricky, "rb").read() + b"@broken\nACGT\nIIII\n" + open(tricky, "rb").read())
    assert run_cli("--ingest-threads", "4", "--segment-bytes", "4096", "--parse-stats", str(bad2), check=False).returncode == 1


def synth_genome(seed, n, plant=None, at=0):
    rng = np.random.default_rng(seed)
    g = H.random_dna(rng, n)
    if plant:
        g = g[:at] + plant + g[at + len(plant):]
    return g


@pytest.mark.gpu
def test_config1_build_and_classify(tmp_path, refdata):
    """BASELINE configs[0]: testData/testQueries.fasta vs a 64-bin IBF.  No E. coli genome exists offline: a seeded
    6.3 Mbp genome is used, with the query's first kilobase planted."""
    (qid, qseq), = H.read_fasta(os.path.join(refdata, "testQueries.fasta"))
    assert len(qseq) == 1890
    genome = synth_genome(1, 6_300_000, plant=qseq[:1000], at=2_345_678)
    ref = tmp_path / "ecoli_like.fasta"
    with open(ref, "w") as fh:
        fh.write(">chr simulated\n")
        for i in range(0, len(genome), 70):
            fh.write(genome[i:i + 70] + "\n")
    out = tmp_path / "RB_out"
    cfg = tmp_path / "build.toml"
    write_config(cfg, "build", out, kmer_size=13, fragment_size=100000, target_files=[ref])
    built = run_cli("--config", str(cfg), "--placement-tries", "1")
    ibf = out / "ecoli_like.ibf"
    assert ibf.exists()
    # where the build's time went (profiles/cli_build.py reads this line at genome scale): every stage named, the totals consistent
    phases = [l for l in built.stdout.splitlines() if l.startswith("BUILD_PHASES")]
    assert len(phases) == 1
    kv = dict(x.split("=", 1) for x in phases[0].split()[1:])
    assert int(kv["bins"]) == 63 and int(kv["bases"]) == 6_299_999
    stages = [float(kv[k]) for k in ("load_seq_s", "alloc_filter_s", "concat_s", "insert_s", "save_s")]
    assert all(x >= 0.0 for x in stages) and float(kv["parse_s"]) > 0.0 and sum(stages) <= float(kv["create_filter_s"]) * 1.001
    # byte-identical to the oracle's restatement of create_filter + store
    o = H.build_filter_like_reference([genome], k=13, fragment_length=100000)
    assert o.n_bins == 63 and o.bin_width == 1  # 6 299 999 bases after cutOutNNNs' dropped base: 62 + 1 bins
    op = tmp_path / "oracle.ibf"
    o.store(str(op))
    assert open(ibf, "rb").read() == open(op, "rb").read()

    # classify usage, deplete mode against the stored IBF; reads: the query, a random read, a short read
    rng = np.random.default_rng(2)
    reads = [(qid, qseq), ("random", H.random_dna(rng, 2000)), ("short", H.random_dna(rng, 200)),
             ("planted_later", H.random_dna(rng, 400) + genome[4_000_000:4_000_800])]
    rf = tmp_path / "reads.fasta"
    with open(rf, "w") as fh:
        for n, s in reads:
            fh.write(">%s\n%s\n" % (n, s))
    for (chunk, maxc) in ((360, 1), (360, 5), (250, 5)):
        write_config(cfg, "classify", out, kmer_size=13, deplete_files=[ibf], read_files=[rf], chunk_length=chunk,
                     max_chunks=maxc)
        # the third variant goes through the multi-device pool (device 0 listed twice on a one-GPU box)
        extra = ["--devices", "0,0", "--batch-reads", "3"] if maxc == 5 and chunk == 250 else []
        stdout = run_cli("--config", str(cfg), *extra).stdout
        exp = dict(found=0, failed=0, too_short=0)
        unclassified = []
        for n, s in reads:
            r = po.classify_read_chunks([o], [], s, chunk, maxc)
            exp["found"] += r["classified"]
            exp["failed"] += r["status"] != po.OK
            exp["too_short"] += r["too_short"]
            if not r["classified"] and not r["too_short"] and r["status"] == po.OK:
                unclassified.append(n)
        line = [l for l in stdout.splitlines() if l.startswith("RESULT")][0]
        assert line == "RESULT found=%d failed=%d too_short=%d readCounter=%d" % (exp["found"], exp["failed"], exp["too_short"], len(reads)), (chunk, maxc, stdout)
        got_un = [n for n, _ in H.read_fasta(str(out / "unclassified.fasta"))]
        assert got_un == unclassified
    assert exp["found"] >= 1  # the planted query is found
    # createLog echo (configReader.cpp:98-200) and the run log
    echo = (out / "configLog.toml").read_text()
    assert "[build]" in echo and "[classify]" in echo and "kmer-size = 13" in echo and "chunk_length = 250" in echo
    assert "classified" in (out / "logs" / "ReadBouncerLog.txt").read_text()


@pytest.mark.gpu
def test_classify_with_fasta_targets_and_deplete(tmp_path, refdata):
    """deplete + target given as FASTA (built on the fly, ibfbuild.hpp:106-123), FASTQ reads with CRLF:
    the reference's classifyTests inputs; per-target tallies and outputs against the oracle driver."""
    tgt_fa = os.path.join(refdata, "classifyTests_test.fasta")
    rng = np.random.default_rng(9)
    dep_seq = H.random_dna(rng, 30000)
    dep_fa = tmp_path / "host.fasta"
    dep_fa.write_text(">host\n" + dep_seq + "\n")
    reads_fq = os.path.join(refdata, "classifyTests_test.fastq")
    extra = tmp_path / "extra.fasta"
    extra.write_text(">hostread\n" + dep_seq[5000:6200] + "\n>noise\n" + H.random_dna(rng, 1500) + "\n")
    out = tmp_path / "out"
    cfg = tmp_path / "c.toml"
    write_config(cfg, "classify", out, kmer_size=13, fragment_size=100000, target_files=[tgt_fa], deplete_files=[dep_fa],
                 read_files=[reads_fq, extra], chunk_length=250, max_chunks=5)
    stdout = run_cli("--config", str(cfg)).stdout
    ot = H.build_filter_like_reference([s for _, s in H.read_fasta(tgt_fa)], k=13)
    od = H.build_filter_like_reference([dep_seq], k=13)
    results = [l for l in stdout.splitlines() if l.startswith("RESULT")]
    assert len(results) == 2
    for line, recs in zip(results, (H.read_fastq(reads_fq), H.read_fasta(str(extra)))):
        found = failed = too_short = 0
        for _, s in recs:
            r = po.classify_read_chunks([od], [ot], s, 250, 5)
            found += r["classified"]; failed += r["status"] != po.OK; too_short += r["too_short"]
        assert line == "RESULT found=%d failed=%d too_short=%d readCounter=%d" % (found, failed, too_short, len(recs))
    assert results[0].startswith("RESULT found=3 ")  # classifygtests.hpp:74-77: 3 of 3 target reads
    assert (out / "classifyTests_test.ibf").exists() and (out / "host.ibf").exists()
    # outputs of the LAST read file (extra.fasta): nothing hits the target, the host read is neither target nor
    # unclassified-by-failure: both records land in unclassified.fasta (classify.hpp:300-301)
    assert H.read_fasta(str(out / "classifyTests_test.fasta")) == []
    assert [n for n, _ in H.read_fasta(str(out / "unclassified.fasta"))] == ["hostread", "noise"]


@pytest.mark.gpu
def test_cpp_mirror_reads_like_the_reference_tests(tmp_path, refdata):
    """tests/cpp/test_mirror.cpp: the reference's ReadTest / IBFTest expectations through include/readbouncer_amd.hpp"""
    exe = os.path.join(ROOT, "readbouncer_amd", "test_mirror")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    p = subprocess.run([exe, refdata, str(tmp_path / "work")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "failures: 0" in p.stdout
    # the other candidate of the reverse-complement-of-N rule, for a whole process (interleave::set_revcomp_of_n ->
    # rb_set_default_revcomp_of_n; a call, not an environment variable): exactly the one expectation that tells the two rules
    # apart fails (30 shared 13-mers on the reverse strand become 18 -> count 0)
    q = subprocess.run([exe, refdata, str(tmp_path / "work4"), "4"], capture_output=True, text=True)
    assert q.returncode != 0 and "failures: 1" in q.stdout and "pn.first" in q.stderr, q.stdout + q.stderr
    # ... and the environment no longer reaches it: RB_REVCOMP_OF_N is not read by anything
    r = subprocess.run([exe, refdata, str(tmp_path / "work5")], capture_output=True, text=True, env=dict(os.environ, RB_REVCOMP_OF_N="4"))
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_pipeline_settings_do_not_change_the_outputs(tmp_path):
    """parser threads, segment size and GPU batch size only change the schedule: RESULT line and every output file are
    byte-identical, in read order, and equal to the oracle's chunk driver -- with reads that are only classified on their
    second or third chunk (the chunks the main thread gathers itself) and reads too short for one chunk"""
    rng = np.random.default_rng(33)
    dep_seq, tgt_seq = H.random_dna(rng, 40000), H.random_dna(rng, 30000)
    (tmp_path / "dep.fasta").write_text(">dep\n" + dep_seq + "\n")
    (tmp_path / "tgt.fasta").write_text(">tgt\n" + tgt_seq + "\n")
    reads = []
    for i in range(3000):
        kind = i % 6
        L = int(rng.integers(300, 1100))
        if kind == 0:
            s = H.random_dna(rng, L)                                             # unclassified
        elif kind == 1:
            p = int(rng.integers(0, len(dep_seq) - L)); s = H.mutate(rng, dep_seq[p:p + L], 0.05)   # deplete, chunk 0
        elif kind == 2:
            p = int(rng.integers(0, len(tgt_seq) - L)); s = H.mutate(rng, tgt_seq[p:p + L], 0.05)   # target, chunk 0
        elif kind == 3:
            p = int(rng.integers(0, len(dep_seq) - 600)); s = H.random_dna(rng, 250) + dep_seq[p:p + 600]   # chunk 1
        elif kind == 4:
            p = int(rng.integers(0, len(tgt_seq) - 400)); s = H.random_dna(rng, 500) + tgt_seq[p:p + 400]   # chunk 2
        else:
            s = H.random_dna(rng, int(rng.integers(1, 250)))                     # too short
        reads.append(("read%d some description" % i, s))
    fq = tmp_path / "reads.fastq"
    with open(fq, "w") as fh:
        for n, s in reads:
            fh.write("@%s\n%s\n+\n%s\n" % (n, s, "".join(rng.choice(list("@>+I5"), size=len(s)))))
    outputs = []
    for tag, extra in (("a", ("--ingest-threads", "1", "--batch-reads", "1000000")),
                       ("b", ("--ingest-threads", "4", "--segment-bytes", "150000", "--batch-reads", "777")),
                       ("c", ("--ingest-threads", "3", "--segment-bytes", "20000", "--batch-reads", "64")),
                       # one and many classifier threads, outputs through shared mappings of the files, engines calibrated first
                       ("d", ("--classify-threads", "1", "--segment-bytes", "90000", "--batch-reads", "5000")),
                       ("e", ("--classify-threads", "7", "--segment-bytes", "30000", "--mmap-output", "--calibrate"))):
        out = tmp_path / ("out_" + tag)
        cfg = tmp_path / (tag + ".toml")
        write_config(cfg, "classify", out, kmer_size=13, fragment_size=1000, deplete_files=[tmp_path / "dep.fasta"],
                     target_files=[tmp_path / "tgt.fasta"], read_files=[fq], chunk_length=250, max_chunks=3)
        stdout = run_cli("--config", str(cfg), *extra).stdout
        line = [l for l in stdout.splitlines() if l.startswith("RESULT")][0]
        files = {p.name: p.read_bytes() for p in sorted(out.glob("*.fasta"))}
        outputs.append((line, files))
    assert all(o == outputs[0] for o in outputs[1:])
    line, files = outputs[0]
    # the oracle's chunk driver on filters built like the reference builds them
    od = H.build_filter_like_reference([dep_seq], k=13, fragment_length=1000)
    ot = H.build_filter_like_reference([tgt_seq], k=13, fragment_length=1000)
    exp_un, exp_t, found, short, failed = [], [], 0, 0, 0
    for n, s in reads:
        r = po.classify_read_chunks([od], [ot], s, 250, 3)
        short += r["too_short"]
        failed += r["status"] != po.OK
        found += r["classified"]
        if r["too_short"] or r["status"] != po.OK:
            continue
        if not r["classified"]:
            exp_un.append((n.split(" ")[0], s))
        elif r["best_target"] >= 0:
            exp_t.append((n.split(" ")[0], s))
    assert line == "RESULT found=%d failed=%d too_short=%d readCounter=%d" % (found, failed, short, len(reads))
    got_un = [(n.split(" ")[0], s) for n, s in H.read_fasta(str(tmp_path / "out_a" / "unclassified.fasta"))]
    got_t = [(n.split(" ")[0], s) for n, s in H.read_fasta(str(tmp_path / "out_a" / "tgt.fasta"))]
    assert got_un == exp_un and got_t == exp_t
    # with both filter sets only target reads count as classified (classify.hpp:58-111): kinds 2 and 4
    assert 950 < found < 1100 and len(exp_t) == found and short == 500 and len(exp_un) > 1000
    assert failed > 50  # a last chunk shorter than k throws in the reference (classify.hpp:306-316): counted, not written


@pytest.mark.gpu
def test_cli_with_one_deplete_and_three_targets_uses_the_merged_table(tmp_path):
    """The reference's README shape through the CLI: 1 deplete + 3 target FASTA files built with ONE fragment_size have the same
    noOfBlocks (IBFBuild.cpp:404-413), so the engine serves them from a merged table when the batch is large and from the
    latency kernels when it is small -- RESULT line, per-target FASTA files and unclassified.fasta are byte-identical either
    way and equal to the oracle's chunk driver, credited target included."""
    rng = np.random.default_rng(71)
    genomes = {"dep": H.random_dna(rng, 100000), "ta": H.random_dna(rng, 40000), "tb": H.random_dna(rng, 25000),
               "tc": H.random_dna(rng, 45000)}
    for name, seq in genomes.items():
        (tmp_path / (name + ".fasta")).write_text(">" + name + "\n" + seq + "\n")
    names = list(genomes)
    reads = []
    for i in range(6000):
        kind = i % 6
        L = int(rng.integers(260, 700))
        if kind < 4:
            g = genomes[names[kind]]
            p = int(rng.integers(0, len(g) - L))
            s = H.mutate(rng, g[p:p + L], 0.06)
        elif kind == 4:
            s = H.random_dna(rng, L)
        else:  # chimera: target on the first chunk, deplete on the second
            a, b = genomes["tb"], genomes["dep"]
            pa, pb = int(rng.integers(0, len(a) - 250)), int(rng.integers(0, len(b) - 300))
            s = a[pa:pa + 250] + b[pb:pb + 300]
        reads.append(("r%d" % i, s))
    fq = tmp_path / "reads.fastq"
    with open(fq, "w") as fh:
        for n, s in reads:
            fh.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    outputs = []
    for tag, batch in (("big", "100000"), ("small", "64")):
        out = tmp_path / ("out_" + tag)
        cfg = tmp_path / (tag + ".toml")
        write_config(cfg, "classify", out, kmer_size=13, fragment_size=1000, deplete_files=[tmp_path / "dep.fasta"],
                     target_files=[tmp_path / "ta.fasta", tmp_path / "tb.fasta", tmp_path / "tc.fasta"], read_files=[fq],
                     chunk_length=250, max_chunks=2)
        stdout = run_cli("--config", str(cfg), "--batch-reads", batch).stdout
        line = [l for l in stdout.splitlines() if l.startswith("RESULT")][0]
        files = {p.name: p.read_bytes() for p in sorted(out.glob("*.fasta"))}
        outputs.append((line, files))
    assert outputs[0] == outputs[1]
    line, files = outputs[0]
    views = {k: H.build_filter_like_reference([v], k=13, fragment_length=1000) for k, v in genomes.items()}
    assert len({v.n_blocks for v in views.values()}) == 1  # one fragment_size -> one noOfBlocks: what the merged table needs
    assert [v.bin_width for v in views.values()] == [2, 1, 1, 1]
    targets = [views["ta"], views["tb"], views["tc"]]
    found, per_target = 0, {0: [], 1: [], 2: []}
    for n, s in reads:
        r = po.classify_read_chunks([views["dep"]], targets, s, 250, 2)
        assert r["status"] == po.OK and not r["too_short"]
        found += r["classified"]
        if r["classified"]:
            per_target[r["best_target"]].append(n)
    assert line == "RESULT found=%d failed=0 too_short=0 readCounter=%d" % (found, len(reads))
    for i, tname in enumerate(("ta", "tb", "tc")):
        got = [n for n, _ in H.read_fasta(str(tmp_path / "out_big" / (tname + ".fasta")))]
        assert got == per_target[i], tname
        assert len(got) > 700
    assert 3700 < found < 4100  # the three target kinds + the chimeras whose first chunk is target 'tb'


@pytest.mark.gpu
def test_verify_ibf_first_contact_check(tmp_path):
    """--verify-ibf: re-inserting the reference a filter was built from must set no new bit.  A file written by
    the oracle builder passes; the same sequences hashed under another seedValue -- a stand-in for "the recalled SeqAn
    constants are wrong" -- fail with about (1 - load) of the re-inserted bits on clear positions; the wrong FASTA or
    fragment size is told apart by the bin count."""
    rng = np.random.default_rng(4242)
    seqs = [H.random_dna(rng, 250_000), H.random_dna(rng, 90_000) + "N" * 50 + H.random_dna(rng, 30_000), "ACGT"]
    fasta = tmp_path / "ref.fasta"
    fasta.write_text("".join(">s%d some description\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    good = H.build_filter_like_reference(seqs, k=13, fragment_length=100000)
    good.store(str(tmp_path / "good.ibf"))
    p = run_cli("--verify-ibf", str(tmp_path / "good.ibf"), "--reference", str(fasta))
    assert "VERIFY OK" in p.stdout and " new_bits=0 " in p.stdout and "explained=1 " in p.stdout, p.stdout
    # the same build under a different seed
    cleaned = [po.cut_out_nnns(s) for s in seqs if len(s) >= 13]
    n_bins = sum(len(c) // 100000 + 1 for c in cleaned)
    bad = po.OracleIBF(n_bins, 3, 13, po.calculate_filter_size_bits(100000, 13, 3, 0.01, n_bins))
    bad.set_seed_for_tests(0x9E3779B97F4A7C15)
    b = 0
    for c in cleaned:
        b = bad.add_sequence(po.encode(c), 100000, b)
    bad.store(str(tmp_path / "bad.ibf"))
    p = run_cli("--verify-ibf", str(tmp_path / "bad.ibf"), "--reference", str(fasta), check=False)
    assert p.returncode == 3 and "VERIFY FAILED" in p.stdout and "constants" in p.stdout, p.stdout
    fields = dict(kv.split("=") for kv in p.stdout.split("VERIFY file=")[1].split("\n")[0].split()[1:])
    assert int(fields["new_bits"]) > 0.7 * int(fields["rebuilt_bits"])  # about (1 - load) of them
    # right constants, wrong fragment size: told apart by the bin count
    p = run_cli("--verify-ibf", str(tmp_path / "good.ibf"), "--reference", str(fasta), "--fragment-size", "50000", check=False)
    assert p.returncode == 3 and "wrong FASTA or fragment size" in p.stdout
    # a file whose metadata tail holds values the reference never writes still loads, with a warning
    odd = po.OracleIBF(n_bins, 2, 13, good.n_bits)
    odd.store(str(tmp_path / "odd.ibf"))
    p = run_cli("--verify-ibf", str(tmp_path / "odd.ibf"), "--reference", str(fasta), check=False)
    assert "WARNING" in p.stdout and "noOfHashFunc = 2" in p.stdout


@pytest.mark.gpu
def test_unclassified_fasta_is_written_like_seqan_writes_it(tmp_path, refdata):
    """classify.hpp:301: seqan::writeRecord(UnclassifiedOut, id, (seqan::Dna5String)seq) -- Dna5 alphabet (upper case,
    U -> T, every other non-ACGT byte -> N), SeqAn's default 70-column FASTA lines.  Target FASTAs get the raw read
    on one line (classify.hpp:288-289)."""
    tgt_fa = os.path.join(refdata, "classifyTests_test.fasta")
    tgt_seq = H.read_fasta(tgt_fa)[0][1]
    rng = np.random.default_rng(31)
    raw = H.random_dna(rng, 333)
    mixed = raw[:100].lower() + "RYKM-*" + raw[100:200] + "uU" + raw[200:]        # 341 bases
    hit = tgt_seq[100:400].lower()[:150] + tgt_seq[250:400]                       # target read, partly lower case
    reads = tmp_path / "reads.fasta"
    reads.write_text(">mixed case and IUPAC\n%s\n>exact70\n%s\n>hit\n%s\n" % (mixed, raw[:280], hit))
    out = tmp_path / "out"
    cfg = tmp_path / "c.toml"
    write_config(cfg, "classify", out, kmer_size=13, fragment_size=100000, target_files=[tgt_fa], read_files=[reads],
                 chunk_length=250, max_chunks=1)
    run_cli("--config", str(cfg))
    text = (out / "unclassified.fasta").read_text().split("\n")
    exp_mixed = raw[:100] + "NNNNNN" + raw[100:200] + "TT" + raw[200:]
    assert text[0] == ">mixed case and IUPAC"
    assert text[1:6] == [exp_mixed[i:i + 70] for i in range(0, 341, 70)]
    assert text[6] == ">exact70" and text[7:11] == [raw[i:i + 70] for i in range(0, 280, 70)] and text[11:] == [""]
    assert (out / "classifyTests_test.fasta").read_text() == ">hit\n%s\n" % hit


def test_ingest_worker_failures_do_not_terminate():
    """tests/cpp/test_seqio.cpp (CPU only): refused page-locked blocks fall back to the heap, an exception inside a parser
    thread ends the stream with an error segment"""
    exe = os.path.join(ROOT, "readbouncer_amd", "test_seqio")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr


def test_pool_work_queue_on_cpu():
    """tests/cpp/test_workq.cpp (CPU only): the host-thread machinery of the one-process pool -- per-worker FIFOs, jobs, the
    least-loaded pick -- with six client threads on four workers: every part runs once, errors reach their caller, calls of
    different clients overlap, shutdown drains the queues"""
    exe = os.path.join(ROOT, "readbouncer_amd", "test_workq")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "failures: 0" in p.stdout, p.stdout + p.stderr


def test_filter_load_reader_threads_on_cpu(tmp_path):
    """tests/cpp/test_io.cpp (CPU only): csrc/rb_io.h -- one gang of reader threads per filter load serving every chunk, a request
    beyond the end of the file and a file that shrinks while it is read end in `false` (the other parts stop at their next 4 MiB
    piece), gangs of one and of more threads than parts, an invalid descriptor"""
    exe = os.path.join(ROOT, "readbouncer_amd", "test_io")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    p = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "rb_io: ok" in p.stdout, p.stdout + p.stderr


def test_live_step_host_logic_on_cpu():
    """tests/cpp/test_live.cpp (CPU only): csrc/rb_live.cpp with the engine call replaced by a stand-in at link time -- micro-batches
    of every size (ids repeated within a batch, short reads, reads that stay undecided past the 1 500 bp cut-off) give the actions of
    the reference's chunk-by-chunk loop (adaptive_sampling.hpp:227-350); a call that fails as a whole changes nothing; four threads
    on one handle.  The same binary runs under ASan / UBSan / TSan in profiles/sanitize_cpu.sh."""
    exe = os.path.join(ROOT, "readbouncer_amd", "test_live")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "failures: 0" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_usage_target_replays_chunks_through_the_live_step(tmp_path):
    """usage = "target" on the TOML surface (main.cpp:365-378) as an offline replay: pre-basecalled chunks in arrival order
    -> rb_live_process micro-batches -> action list, equal to the sequential restatement of classify_live_reads
    (adaptive_sampling.hpp:227-350) driven by the oracle, for every micro-batch size."""
    from tests.test_gpu_live import reference_live
    rng = np.random.default_rng(77)
    host, bug = H.random_dna(rng, 30000), H.random_dna(rng, 30000)
    (tmp_path / "host.fasta").write_text(">host\n%s\n" % host)
    (tmp_path / "bug.fasta").write_text(">bug\n%s\n" % bug)
    stream = []
    mols = {}
    for m in range(120):
        kind, L = m % 5, int(rng.integers(300, 2600))
        src = host if kind in (0, 4) else bug
        s = int(rng.integers(0, 30000 - L))
        mol = (H.mutate(rng, src[s:s + L], 0.25 if kind == 4 else 0.12) if kind in (0, 1, 4)
               else H.random_dna(rng, L) if kind == 3 else H.mutate(rng, host[s:s + L // 2] + bug[s:s + L // 2], 0.05))
        chunks, pos = [], 0
        while pos < len(mol):
            step = int(rng.choice([8, 120, 250, 360, 400]))
            chunks.append(mol[pos:pos + step]); pos += step
        mols["read%03d" % m] = chunks
    alive = sorted(mols)
    while alive:  # interleave the reads, chunks of one read stay in order
        rid = alive[int(rng.integers(0, len(alive)))]
        stream.append((rid, mols[rid].pop(0)))
        if not mols[rid]:
            alive.remove(rid)
    fq = tmp_path / "chunks.fastq"
    fq.write_text("".join("@%s ch=%d chunk=%d\n%s\n+\n%s\n" % (rid, i % 512, i, s, "I" * len(s)) for i, (rid, s) in enumerate(stream)))
    od = H.build_filter_like_reference([host], k=13)
    ot = H.build_filter_like_reference([bug], k=13)
    exp, exp_once = reference_live([od], [ot], stream, r=0.1)
    names = {0: "none", 1: "unblock_read", 2: "stop_receiving_data"}
    for live_batch in (1, 7, 64):
        out = tmp_path / ("out%d" % live_batch)
        cfg = tmp_path / ("t%d.toml" % live_batch)
        cfg.write_text('usage = "target"\noutput_directory = \'%s\'\nlog_directory = \'%s/logs\'\n\n[IBF]\nkmer_size = 13\n'
                       "fragment_size = 100000\nexp_seq_error_rate = 0.1\ndeplete_files = ['%s']\ntarget_files = ['%s']\n"
                       "read_files = ['%s']\n\n[MinKNOW]\nhost = \"localhost\"\nport = \"9502\"\nflowcell = \"MS00000\"\n\n"
                       "[Basecaller]\ncaller = \"DeepNano\"\n" % (out, out, tmp_path / "host.fasta", tmp_path / "bug.fasta", fq))
        p = run_cli("--config", str(cfg), "--live-batch", str(live_batch))
        rows = [l.split("\t") for l in (out / "live_actions.tsv").read_text().splitlines()[1:]]
        assert [r[1] for r in rows] == [rid for rid, _ in stream]
        assert [(r[2], int(r[3])) for r in rows] == [(names[a], st) for a, st in exp]
        summary = [l for l in p.stdout.splitlines() if l.startswith("LIVE ")][0]
        assert "chunks=%d " % len(stream) in summary and "pending=%d " % len(exp_once) in summary
        assert "unblock=%d " % sum(a == 1 for a, _ in exp) in summary and "stop=%d " % sum(a == 2 for a, _ in exp) in summary
    assert sum(a == 1 for a, _ in exp) > 5 and sum(a == 2 for a, _ in exp) > 5
    # without read_files the live connection would be needed: refused, with a pointer to the replay
    cfg = tmp_path / "nofiles.toml"
    cfg.write_text('usage = "target"\noutput_directory = \'%s\'\nlog_directory = \'%s/logs\'\n\n[IBF]\ndeplete_files = [\'%s\']\n'
                   % (tmp_path / "o", tmp_path / "o", tmp_path / "host.fasta"))
    p = run_cli("--config", str(cfg), check=False)
    assert p.returncode == 2 and "read_files" in p.stderr
