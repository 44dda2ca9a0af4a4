#!/usr/bin/env python3
"""Generate tests/golden/*.json by RUNNING THE REFERENCE (build container only).

Usage (from the repo root, in the container that has /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference (Zhijian-Mei/py-debruijn, ``debruijn.py`` + ``getScore`` of
``II_assembleFromReads.py``) holds no tests or golden vectors of its own
(SURVEY.md section 4), so parity is pinned by executing it here on fixed inputs and
committing inputs + outputs as data.  Nothing of the reference's text is copied;
the fixtures hold reads and results only.  The reference never travels to the
GPU box: tests read the JSON, not /root/reference.

Case families: the peptide example at
II_assembleFromReads.py:55; synthetic DNA (k in {5, 21, 31}, error 0 % / 1 %,
threshold in {1, 2, 3}, both ``final`` modes); a cycle; reads with len <= k;
duplicate reads; a multi-k driver run (II_assembleFromReads.py:56-75).

Large cases store a sha256 digest of the canonical result plus summary counts
instead of the full result ("digest" cases); small cases store everything.
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("DBG_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
import importlib.util  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# The reference is loaded BY PATH (this repo ships a drop-in that is also called debruijn.py).  Its driver does
# ``import debruijn as db``: the name must resolve to the reference while the driver module is loaded.
ref = _load("debruijn", os.path.join(REF, "debruijn.py"))
assert os.path.realpath(ref.__file__).startswith(os.path.realpath(REF)), "fixtures must come from the reference"
synth = _load("synth", os.path.join(ROOT, "py-debruijn_amd", "synth.py"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def canonical(g, pull, branch, pulled, ect, contigs, stdout):
    V, E = g
    return {
        "vertices": [[v, V[v].indegree, V[v].outdegree] for v in V],
        "edges": [[v, list(E[v])] for v in E],
        "pull_out_read": list(pull),
        "branch_kmer": list(branch),
        "already_pull_out": list(pulled),
        "edge_count_table": [[n, c] for n, c in ect.items()],
        "contigs": list(contigs),
        "stdout": stdout,
    }


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def run_reference(reads, k, threshold, final):
    """construct_graph + output_contigs of the reference, big stack for the DFS."""
    box = {}

    def work():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            g, pull, branch, pulled, ect = ref.construct_graph(list(reads), k, threshold=threshold, final=final)
            contigs = ref.output_contigs(g, branch, pulled)
        box["res"] = canonical(g, pull, branch, pulled, ect, contigs, buf.getvalue())

    sys.setrecursionlimit(400000)
    threading.stack_size(512 * 1024 * 1024)
    t = threading.Thread(target=work)
    t.start()
    t.join()
    return box["res"]


def run_reference_driver(reads, kl, ku, threshold):
    """II_assembleFromReads.py:56-75 around the imported reference functions."""
    box = {}

    def work():
        spec = importlib.util.spec_from_file_location("ref_II", os.path.join(REF, "II_assembleFromReads.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)  # its "import debruijn" finds the reference in sys.modules
        assert mod.db is ref
        sequences = list(reads)
        trace = {}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            for k in range(kl, ku + 1):
                if k <= ku - 1:
                    g, pull, branch, pulled, ect = ref.construct_graph(sequences, k, threshold=threshold)
                else:
                    g, pull, branch, pulled, ect = ref.construct_graph(sequences, k, threshold=threshold, final=True)
                sequences = ref.output_contigs(g, branch, pulled)
                sequences.sort(key=lambda x: mod.getScore(ect, x, k), reverse=True)
                if k == ku:
                    break
                trace[str(k)] = {"contigs": list(sequences), "pull_out_read": list(pull)}
                sequences.extend(pull)
        box["res"] = {"final_contigs": list(sequences), "trace": trace}

    sys.setrecursionlimit(400000)
    threading.stack_size(512 * 1024 * 1024)
    t = threading.Thread(target=work)
    t.start()
    t.join()
    return box["res"]


def summary(res):
    return {
        "n_vertices": len(res["vertices"]),
        "n_edge_names": len(res["edge_count_table"]),
        "n_branch": len(res["branch_kmer"]),
        "n_pulled": len(res["already_pull_out"]),
        "n_pull_reads": len(res["pull_out_read"]),
        "n_contigs": len(res["contigs"]),
        "sum_edge_counts": sum(c for _, c in res["edge_count_table"]),
        "n_indeg0": sum(1 for _, i, _ in res["vertices"] if i == 0),
    }


def part_digests(res):
    """Per-field digests, both order-sensitive and order-independent."""
    d = {}
    for key in ("vertices", "edges", "pull_out_read", "branch_kmer", "already_pull_out",
                "edge_count_table", "contigs"):
        d[key] = digest(res[key])
        d[key + "_sorted"] = digest(sorted(res[key], key=lambda x: json.dumps(x)))
    # adjacency with successor lists as sets (the product orders count ties by base)
    d["edges_as_sets"] = digest(sorted([[v, sorted(s)] for v, s in res["edges"]]))
    return d


def emit(name, inputs, res, full):
    case = {"name": name, "inputs": inputs, "summary": summary(res), "digests": part_digests(res)}
    if full:
        case["result"] = res
    with open(os.path.join(GOLDEN, name + ".json"), "w") as fh:
        json.dump(case, fh, separators=(",", ":"))
    print(f"{name}: {case['summary']}")


def dna_case(name, seed, G, n, L, err, k, threshold, final, full):
    reads = synth.reads_list(seed, G, n, L, err)
    res = run_reference(reads, k, threshold, final)
    gen = {"seed": seed, "genome_len": G, "n_reads": n, "read_len": L, "err_rate": err}
    inputs = {"k": k, "threshold": threshold, "final": final, "generator": gen}
    if full:
        inputs["reads"] = reads
    else:
        inputs["reads_checksum"] = synth.checksum(synth.reads_ascii(seed, G, n, L, err))
    emit(name, inputs, res, full)


def main_wide():
    """k in 32..63 (two-word k-mers on the device, BASELINE.json configs[4]): `make_golden.py wide`."""
    os.makedirs(GOLDEN, exist_ok=True)
    dna_case("dna_small_k33_e1_t2", 41, 400, 90, 70, 0.02, 33, 2, False, True)
    dna_case("dna_small_k40_e1_t3_final", 42, 300, 60, 80, 0.02, 40, 3, True, True)
    dna_case("dna_med_k32_e1_t3", 43, 6000, 600, 120, 0.01, 32, 3, False, False)
    dna_case("dna_med_k47_e0_t2", 44, 6000, 600, 120, 0.0, 47, 2, False, False)
    dna_case("dna_med_k63_e1_t2", 45, 8000, 700, 150, 0.01, 63, 2, False, False)
    dna_case("dna_med_k63_e2_t2_final", 46, 3000, 200, 150, 0.003, 63, 2, True, False)
    reads = synth.reads_list(47, 1500, 160, 90, 0.01)
    res = run_reference_driver(reads, 30, 34, 2)  # crosses the one-word / two-word boundary
    with open(os.path.join(GOLDEN, "driver_dna_k30_34.json"), "w") as fh:
        json.dump({"name": "driver_dna_k30_34",
                   "inputs": {"reads": reads, "k_lowerlimit": 30, "k_upperlimit": 34, "threshold": 2},
                   "result": res}, fh, separators=(",", ":"))
    print("driver_dna_k30_34", len(res["final_contigs"]))

    # randomized cases with repeats longer than k (branches), errors near read ends (tips), tandem repeats (cycles)
    import random
    rng = random.Random(20260412)
    fuzz = []
    for i in range(160):
        alpha = rng.choice(["ACGT", "ACGT", "ACG", "AC"])
        k = rng.choice([32, 33, 40, 47, 48, 56, 62, 63, rng.randint(32, 63)])
        rnd = lambda n: "".join(rng.choice(alpha) for _ in range(n))
        kind = rng.random()
        if kind < 0.45:    # a block longer than k occurs twice with different continuations
            R = rnd(k + rng.randint(1, 30))
            G = rnd(rng.randint(5, 40)) + R + rnd(rng.randint(5, 50)) + R + rnd(rng.randint(5, 40))
        elif kind < 0.65:  # tandem repeat: the k-mers close a cycle
            unit = rnd(rng.randint(3, 20))
            G = rnd(rng.randint(0, 10)) + unit * ((k + 40) // len(unit) + 2) + rnd(rng.randint(0, 10))
        else:
            G = rnd(rng.randint(k + 5, k + 120))
        reads = []
        for _ in range(rng.randint(1, 18)):
            L = rng.randint(max(1, k - 3), min(len(G), k + 45))
            st = rng.randint(0, len(G) - L)
            r = list(G[st:st + L])
            if rng.random() < 0.45:   # substitution, often within the last few bases (a tip)
                j = L - 1 - rng.randint(0, 5) if rng.random() < 0.6 else rng.randrange(L)
                r[max(j, 0)] = rng.choice(alpha)
            reads.append("".join(r))
        if rng.random() < 0.3:
            reads += reads[:rng.randint(1, 3)]  # duplicates raise counts: pruning thresholds matter
        thr = rng.choice([1, 2, 2, 3, 3, 5])
        final = rng.random() < 0.4
        res = run_reference(reads, k, thr, final)
        fuzz.append({"inputs": {"reads": reads, "k": k, "threshold": thr, "final": final}, "result": res})
    with open(os.path.join(GOLDEN, "fuzz_wide.json"), "w") as fh:
        json.dump(fuzz, fh, separators=(",", ":"))
    print("fuzz_wide:", len(fuzz), "cases;",
          sum(1 for c in fuzz if c["result"]["already_pull_out"]), "with pulled tips;",
          sum(1 for c in fuzz if c["result"]["branch_kmer"]), "with branches;",
          sum(1 for c in fuzz if c["result"]["contigs"]), "with contigs;",
          sum(1 for c in fuzz if c["result"]["vertices"] and not c["result"]["contigs"]), "with vertices but no contig")


def main_pepwide():
    """Peptides (and other non-ACGT alphabets) with k = 12..40: by-reference tables on the device: `make_golden.py pepwide`."""
    import random
    rng = random.Random(20260413)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    fuzz = []
    for i in range(120):
        alpha = rng.choice([aa, aa, aa[:6], "EVQLG", "KR", "ACGTN", "acgtACGT"])
        k = rng.choice([12, 13, 16, 20, 24, 25, 31, 40, rng.randint(12, 40)])
        rnd = lambda n: "".join(rng.choice(alpha) for _ in range(n))
        kind = rng.random()
        if kind < 0.45:    # a block longer than k occurs twice with different continuations
            R = rnd(k + rng.randint(1, 20))
            G = rnd(rng.randint(5, 30)) + R + rnd(rng.randint(5, 40)) + R + rnd(rng.randint(5, 30))
        elif kind < 0.65:  # tandem repeat: the k-mers close a cycle
            unit = rnd(rng.randint(2, 12))
            G = rnd(rng.randint(0, 8)) + unit * ((k + 30) // len(unit) + 2) + rnd(rng.randint(0, 8))
        else:
            G = rnd(rng.randint(k + 5, k + 90))
        reads = []
        for _ in range(rng.randint(1, 16)):
            L = rng.randint(max(1, k - 3), min(len(G), k + 35))
            st = rng.randint(0, len(G) - L)
            r = list(G[st:st + L])
            if rng.random() < 0.45:
                j = L - 1 - rng.randint(0, 5) if rng.random() < 0.6 else rng.randrange(L)
                r[max(j, 0)] = rng.choice(alpha)
            reads.append("".join(r))
        if rng.random() < 0.3:
            reads += reads[:rng.randint(1, 3)]
        thr = rng.choice([1, 2, 2, 3, 3, 5])
        final = rng.random() < 0.4
        res = run_reference(reads, k, thr, final)
        fuzz.append({"inputs": {"reads": reads, "k": k, "threshold": thr, "final": final}, "result": res})
    with open(os.path.join(GOLDEN, "fuzz_peptide_wide.json"), "w") as fh:
        json.dump(fuzz, fh, separators=(",", ":"))
    print("fuzz_peptide_wide:", len(fuzz), "cases;",
          sum(1 for c in fuzz if c["result"]["already_pull_out"]), "with pulled tips;",
          sum(1 for c in fuzz if c["result"]["branch_kmer"]), "with branches;",
          sum(1 for c in fuzz if c["result"]["contigs"]), "with contigs")
    pep = ['EVQLVESGGGLVQPGGSLRLSCAAS', 'GGGLVQPGGSLRLSCAASGFTFS', 'LVQPGGSLRLSCAASGFNIKDTYIH', 'EVQLVESGGGLVQPGGSLRL',
           'SLRLSCAASGFNIKDTYIHWVRQAPGK', 'GGSLRLSCAASGFNIKDTYIHWV']
    res = run_reference_driver(pep, 10, 14, 2)  # crosses the packed / by-reference boundary (k = 11 -> 12)
    with open(os.path.join(GOLDEN, "driver_peptide_k10_14.json"), "w") as fh:
        json.dump({"name": "driver_peptide_k10_14",
                   "inputs": {"reads": pep, "k_lowerlimit": 10, "k_upperlimit": 14, "threshold": 2},
                   "result": res}, fh, separators=(",", ":"))
    print("driver_peptide_k10_14", len(res["final_contigs"]))


def main_aux():
    """The two helpers of debruijn.py the pipeline does not call (get_kmers :35-75, get_graph_from_kmers :78-95):
    `make_golden.py aux`."""
    import random
    rng = random.Random(20260414)
    cases = []
    for i in range(300):
        alpha = rng.choice(["ACGT", "AC", "EVQLG", "ACDEFGHIKLMNPQRSTVWY"])
        k = rng.randint(2, 7)
        n = rng.randint(0, 12)
        G = "".join(rng.choice(alpha) for _ in range(rng.randint(6, 40)))
        seqs = []
        for _ in range(n):
            if rng.random() < 0.7:
                L = rng.randint(1, min(len(G), k + 8))
                st = rng.randint(0, len(G) - L)
                seqs.append(G[st:st + L])
            else:
                seqs.append("".join(rng.choice(alpha) for _ in range(rng.randint(0, k + 6))))
        if rng.random() < 0.3 and seqs:
            seqs += [rng.choice(seqs) for _ in range(rng.randint(1, 3))]
        work = list(seqs)
        kmers = ref.get_kmers(work, k)
        V, E = ref.get_graph_from_kmers(list(kmers), k)
        cases.append({"sequences": seqs, "k": k, "kmers": kmers, "sequences_after": work,
                      "vertices": [[v, V[v].indegree, V[v].outdegree] for v in V], "edges": [[v, list(E[v])] for v in E]})
    with open(os.path.join(GOLDEN, "aux_kmers.json"), "w") as fh:
        json.dump(cases, fh, separators=(",", ":"))
    print("aux_kmers:", len(cases), "cases;", sum(1 for c in cases if c["sequences_after"] != c["sequences"]), "with glued or dropped sequences;",
          sum(1 for c in cases if any(i > 1 or o > 1 for _, i, o in c["vertices"])), "with degrees above 1")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    pep = ['EVQLVE', 'QLVAPG', 'LVESGGAL', 'LVESGGGL']  # II_assembleFromReads.py:55 (input only)
    for k in (3, 4):
        for final in (False, True):
            res = run_reference(pep, k, 2, final)
            emit(f"peptide_k{k}_{'final' if final else 'nonfinal'}",
                 {"reads": pep, "k": k, "threshold": 2, "final": final}, res, True)

    # hand-made edge cases
    hand = {
        "cycle": (["ACGTACGTACGTA", "GTACGTAC"], 3),
        "cycle_rho": (["TTACGACGACGAC", "CGACG"], 3),
        "short_reads": (["ACG", "AC", "", "ACGT", "ACGTT", "A", "CGTTA"], 3),
        "len_eq_k_is_branch": (["AAC", "AACG", "AACT", "AACG", "AACT", "TAAC"], 3),
        "duplicates": (["ACGTTGCA", "ACGTTGCA", "ACGTTGCA", "ACGTAGCA", "TTGCAAAC"], 4),
        "homopolymer": (["AAAAAAAAAA", "AAAAACAAAA", "CAAAAAAG"], 4),
        "tips_order": (["GATTACAGATTTCA", "GATTACAGATTTCA", "GATTACAGC", "TTACAGA", "ACAGCT", "CAGATTTG",
                        "CCGATTACAG", "GATTACAGATTTCA"], 4),
        "single_read": (["ACGTACCA"], 2),
    }
    for name, (reads, k) in hand.items():
        for thr in (1, 2, 3):
            for final in (False, True):
                res = run_reference(reads, k, thr, final)
                emit(f"hand_{name}_t{thr}_{'final' if final else 'nonfinal'}",
                     {"reads": reads, "k": k, "threshold": thr, "final": final}, res, True)

    # small synthetic DNA, full results
    dna_case("dna_small_k5_e0_t2", 11, 300, 60, 30, 0.0, 5, 2, False, True)
    dna_case("dna_small_k5_e1_t2", 12, 300, 80, 30, 0.02, 5, 2, False, True)
    dna_case("dna_small_k5_e1_t3_final", 13, 200, 40, 25, 0.02, 5, 3, True, True)
    dna_case("dna_small_k9_e1_t1", 14, 500, 120, 40, 0.02, 9, 1, False, True)
    # medium synthetic DNA, digest only (k = 21 / 31, the BASELINE shapes scaled down)
    dna_case("dna_med_k21_e0_t2", 21, 5000, 600, 100, 0.0, 21, 2, False, False)
    dna_case("dna_med_k21_e1_t2", 22, 5000, 600, 100, 0.01, 21, 2, False, False)
    dna_case("dna_med_k31_e1_t2", 23, 8000, 700, 150, 0.01, 31, 2, False, False)
    dna_case("dna_med_k31_e1_t3", 24, 8000, 700, 150, 0.01, 31, 3, False, False)
    dna_case("dna_med_k31_e2_t2_final", 25, 3000, 200, 150, 0.003, 31, 2, True, False)
    dna_case("dna_med_k16_e1_t2", 26, 4000, 600, 80, 0.01, 16, 2, False, False)

    # multi-k driver (II_assembleFromReads.py:56-75)
    for name, seed, G, n, L, err, kl, ku, thr in (
            ("driver_dna_k5_8", 31, 250, 60, 30, 0.02, 5, 8, 2),
            ("driver_dna_k12_15", 32, 1500, 200, 60, 0.01, 12, 15, 2)):
        reads = synth.reads_list(seed, G, n, L, err)
        res = run_reference_driver(reads, kl, ku, thr)
        case = {"name": name,
                "inputs": {"reads": reads, "k_lowerlimit": kl, "k_upperlimit": ku, "threshold": thr},
                "result": res}
        with open(os.path.join(GOLDEN, name + ".json"), "w") as fh:
            json.dump(case, fh, separators=(",", ":"))
        print(name, len(res["final_contigs"]), {k: len(v["contigs"]) for k, v in res["trace"].items()})
    res = run_reference_driver(pep, 3, 5, 2)
    with open(os.path.join(GOLDEN, "driver_peptide_k3_5.json"), "w") as fh:
        json.dump({"name": "driver_peptide_k3_5",
                   "inputs": {"reads": pep, "k_lowerlimit": 3, "k_upperlimit": 5, "threshold": 2},
                   "result": res}, fh, separators=(",", ":"))

    # randomized small cases over sub-alphabets of ACGT: dense in branches, tips, cycles
    import random
    rng = random.Random(20260410)
    fuzz = []
    for i in range(400):
        alpha = rng.choice(["ACGT", "ACGT", "ACG", "AC", "AT"])
        k = rng.randint(2, 6)
        n = rng.randint(1, 14)
        if rng.random() < 0.5:  # reads sampled from a tiny genome (shared k-mers, tips from errors)
            G = "".join(rng.choice(alpha) for _ in range(rng.randint(8, 40)))
            reads = []
            for _ in range(n):
                L = rng.randint(1, min(len(G), k + 9))
                st = rng.randint(0, len(G) - L)
                r = list(G[st:st + L])
                if rng.random() < 0.4:
                    j = rng.randrange(L)
                    r[j] = rng.choice(alpha)
                reads.append("".join(r))
        else:
            reads = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, k + 8))) for _ in range(n)]
        thr = rng.choice([1, 2, 2, 3, 3, 5])
        final = rng.random() < 0.4
        res = run_reference(reads, k, thr, final)
        fuzz.append({"inputs": {"reads": reads, "k": k, "threshold": thr, "final": final}, "result": res})
    with open(os.path.join(GOLDEN, "fuzz_small.json"), "w") as fh:
        json.dump(fuzz, fh, separators=(",", ":"))
    print("fuzz_small:", len(fuzz), "cases;",
          sum(1 for c in fuzz if c["result"]["already_pull_out"]), "with pulled tips;",
          sum(1 for c in fuzz if c["result"]["branch_kmer"]), "with branches")

    # randomized peptide cases (the reference's real alphabet): 20 amino acids, plus low-complexity
    # sub-alphabets that make branches, tips and cycles dense; k within the 5-bit packing limit
    rng = random.Random(20260411)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    pfuzz = []
    for i in range(240):
        alpha = rng.choice([aa, aa, aa[:6], "EVQLG", "KR", aa[:12]])
        k = rng.randint(2, 7)
        n = rng.randint(1, 16)
        if rng.random() < 0.6:
            G = "".join(rng.choice(alpha) for _ in range(rng.randint(10, 60)))
            reads = []
            for _ in range(n):
                L = rng.randint(1, min(len(G), k + 12))
                st = rng.randint(0, len(G) - L)
                r = list(G[st:st + L])
                if rng.random() < 0.4:
                    r[rng.randrange(L)] = rng.choice(alpha)
                reads.append("".join(r))
        else:
            reads = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, k + 10))) for _ in range(n)]
        thr = rng.choice([1, 2, 2, 3, 3, 5])
        final = rng.random() < 0.4
        res = run_reference(reads, k, thr, final)
        pfuzz.append({"inputs": {"reads": reads, "k": k, "threshold": thr, "final": final}, "result": res})
    with open(os.path.join(GOLDEN, "fuzz_peptide.json"), "w") as fh:
        json.dump(pfuzz, fh, separators=(",", ":"))
    print("fuzz_peptide:", len(pfuzz), "cases;",
          sum(1 for c in pfuzz if c["result"]["already_pull_out"]), "with pulled tips;",
          sum(1 for c in pfuzz if c["result"]["branch_kmer"]), "with branches;",
          sum(1 for c in pfuzz if c["result"]["contigs"]), "with contigs")

    # generator checksum (host twin; the device twin must match, tests/test_synth.py)
    chk = {"seed": 1, "genome_len": 100000, "n_reads": 10000, "read_len": 100}
    out = {"params": chk,
           "checksum_e0": synth.checksum(synth.reads_ascii(1, 100000, 10000, 100, 0.0)),
           "checksum_e1": synth.checksum(synth.reads_ascii(1, 100000, 10000, 100, 0.01)),
           "first_read_e0": synth.reads_list(1, 100000, 1, 100, 0.0)[0],
           "first_read_e1": synth.reads_list(1, 100000, 1, 100, 0.01)[0]}
    with open(os.path.join(GOLDEN, "synth_checksum.json"), "w") as fh:
        json.dump(out, fh)
    print("synth checksum", out["checksum_e0"], out["checksum_e1"])


def main_support():
    """Read-support scores (IV_sortOutputs.py:10-15): findSupportReadScore of the reference, imported by path, on
    random peptide and DNA score tables.  Scores travel as float.hex() so that the sums compare bit for bit."""
    import random
    iv = _load("ref_IV_sortOutputs", os.path.join(REF, "IV_sortOutputs.py"))
    assert os.path.realpath(iv.__file__).startswith(os.path.realpath(REF))
    rng = random.Random(4242)
    cases = []
    for t in range(60):
        alpha = "ACDEFGHIKLMNPQRSTVWY" if t % 3 else "ACGT"
        n_reads = rng.randint(1, 60)
        reads = {}
        while len(reads) < n_reads:
            L = rng.choice([0] * (1 if t % 10 == 0 else 0) + [1, 2, 3, 4, 5, 6, 8, 12, 20, 33])
            r = "".join(rng.choice(alpha) for _ in range(L))
            if r not in reads:
                # the reference's TSV scores are floats; integers and repeated values on purpose, too
                reads[r] = rng.choice([rng.uniform(0, 100), float(rng.randint(1, 99)), rng.randint(1, 50), 0.1, 1e-3, 97.25])
        keys = list(reads)
        contigs = []
        for _ in range(rng.randint(1, 12)):
            parts = []
            for _ in range(rng.randint(0, 6)):
                if keys and rng.random() < 0.7:
                    r = rng.choice(keys)
                    if r and rng.random() < 0.3:   # a truncated copy: must NOT count unless another read matches
                        r = r[:-1] if rng.random() < 0.5 else r[1:]
                    parts.append(r)
                else:
                    parts.append("".join(rng.choice(alpha) for _ in range(rng.randint(0, 9))))
            contigs.append("".join(parts))
        want = [iv.findSupportReadScore(c, reads) for c in contigs]
        cases.append({"reads": keys, "scores": [reads[r].hex() if isinstance(reads[r], float) else reads[r] for r in keys],
                      "contigs": contigs, "want": [w.hex() if isinstance(w, float) else w for w in want]})
    with open(os.path.join(GOLDEN, "support_scores.json"), "w") as fh:
        json.dump(cases, fh, separators=(",", ":"))
    print("support_scores:", len(cases), "cases;", sum(len(c["contigs"]) for c in cases), "contigs;",
          sum(1 for c in cases for w in c["want"] if w not in (0, "0x0.0p+0")), "non-zero scores")


if __name__ == "__main__":
    if sys.argv[1:] == ["support"]:
        main_support()
    elif sys.argv[1:] == ["wide"]:
        main_wide()
    elif sys.argv[1:] == ["pepwide"]:
        main_pepwide()
    elif sys.argv[1:] == ["aux"]:
        main_aux()
    else:
        main()
        main_wide()
        main_pepwide()
        main_aux()
        main_support()
