#!/usr/bin/env python3
"""
gen_golden.py -- produce the parity fixtures under tests/golden/ by running the
REFERENCE's own function bodies on seeded inputs.

Runs in the build container only (needs /root/reference, which never travels to
the GPU box).  The reference is Python 2.7 and cannot be imported as a module
here (py2 print statements, Biopython absent, sklearn.neighbors.kde removed), so
the wanted FunctionDef / ClassDef / Assign nodes are extracted from the
reference text with ``ast`` at run time and exec'd in a namespace that provides
``xrange = range`` and the modules they use.  No reference text is written into
this repository: only inputs and the outputs the reference produced (data).

Third-party arithmetic (scikit-learn, unpinned by the reference) is whatever is
installed here; its version is recorded in tests/golden/MANIFEST.json.

Usage:  python tools/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import ast
import json
import logging
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phamers_amd import synth  # noqa: E402  (seeded input generator, ours)


class _Py2Division(ast.NodeTransformer):
    """The reference is Python 2: ``/`` between integers is floor division there.  Applied (only where asked
    for) to functions whose arithmetic relies on it, so that they run with the semantics they were written for."""

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            node.op = ast.FloorDiv()
        return node


def extract(path, names, namespace, py2_division=False):
    """exec the top-level defs / assignments called ``names`` from ``path``."""
    text = open(path).read()
    try:
        tree = ast.parse(text)
    except SyntaxError:
        # py2-only statements elsewhere in the file (e.g. a print statement in the __main__
        # block): parse just the top-level blocks that define the wanted names.
        lines = text.splitlines(keepends=True)
        starts = [i for i, ln in enumerate(lines) if ln[:1] not in (' ', '\t', '\n', '#', ')', ']', '}')]
        starts.append(len(lines))
        keep = []
        for a, b in zip(starts[:-1], starts[1:]):
            head = lines[a].split('(')[0].split('=')[0].split()
            if head and head[-1].rstrip(':') in names and (head[0] in ('def', 'class') or len(head) == 1):
                keep.append(''.join(lines[a:b]))
        tree = ast.parse(''.join(keep))
    body = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            body.append(node)
        elif isinstance(node, ast.Assign) and any(
                isinstance(t, ast.Name) and t.id in names for t in node.targets):
            body.append(node)
    found = {getattr(n, 'name', None) or n.targets[0].id for n in body}
    missing = set(names) - found
    if missing:
        raise RuntimeError("not found in %s: %s" % (path, sorted(missing)))
    module = ast.Module(body=body, type_ignores=[])
    if py2_division:
        module = ast.fix_missing_locations(_Py2Division().visit(module))
    exec(compile(module, path, 'exec'), namespace)
    return types.SimpleNamespace(**{n: namespace[n] for n in names})


def load_reference(ref):
    import sklearn
    from sklearn.cluster import KMeans
    from sklearn.metrics import silhouette_score
    from sklearn.neighbors import KNeighborsClassifier
    scripts = os.path.join(ref, 'scripts')
    quiet = logging.getLogger('reference')
    quiet.setLevel(logging.ERROR)

    kmer_ns = {'np': np, 'xrange': range, 'logger': quiet}
    kmer = extract(os.path.join(scripts, 'kmer.py'),
                   ['DNA', 'count_string', 'count', 'sequence_to_integers', 'normalize_counts',
                    'kmers', 'extend_mers', 'get_kmer_index'], kmer_ns)

    learn_ns = {'np': np, 'xrange': range, 'logger': quiet, 'KMeans': KMeans,
                'KNeighborsClassifier': KNeighborsClassifier, 'silhouette_score': silhouette_score}
    learning = extract(os.path.join(scripts, 'learning.py'),
                       ['kmeans_seed', 'distances', 'closest_to', 'get_centroids', 'knn', 'kmeans'],
                       learn_ns)

    basic = extract(os.path.join(scripts, 'basic.py'), ['generate_summary'], {'np': np, 'xrange': range})
    id_parser = extract(os.path.join(scripts, 'id_parser.py'), ['get_contig_id'], {'logger': quiet})
    fio_ns = {'np': np, 'xrange': range, 'logger': quiet, 'kmer': kmer, 'basic': basic, 'id_parser': id_parser}
    fileIO = extract(os.path.join(scripts, 'fileIO.py'),
                     ['read_feature_file', 'save_counts', 'save_phamer_scores', 'read_phamer_output'], fio_ns)

    ph_ns = {'np': np, 'xrange': range, 'logger': quiet, 'os': os, 'kmer': kmer,
             'learning': learning, 'fileIO': fileIO,
             '__file__': os.path.join(scripts, 'phamer.py')}
    phamer = extract(os.path.join(scripts, 'phamer.py'), ['phamer_scorer', 'score_points'], ph_ns)
    return kmer, learning, fileIO, phamer, sklearn.__version__


# ---------------------------------------------------------------------------
COUNT_LITERALS = {
    'kat_AAAT': 'AAAT',                       # scripts/kmer.py:89-91 docstring known answer
    'empty': '',
    'len3': 'ATG',
    'len4': 'GATC',
    'len5': 'GATCA',
    'mixed_invalid': 'ATGCATGCNATGCatgcATGC',  # N + lower case break windows
    'all_N': 'N' * 40,
    'N_start': 'NNNN' + 'ATGCGTACGTTAGC' * 3,
    'N_end': 'ATGCGTACGTTAGC' * 3 + 'NNN',
    'N_middle_runs': 'ATGCGT' + 'N' + 'ACGTTAGCAT' + 'NN' + 'GCGCGCAATT' + 'NNNNN' + 'TTGACA',
    'iupac': 'ATGCRYKMSWBDHVNATGCATGC',
    'lowercase': 'atgcatgcatgcatgc',
    'homopolymer_A': 'A' * 100,
    'homopolymer_C': 'C' * 257,
    'dinuc_repeat': 'AT' * 300,
    'digits_and_dash': 'ATGC-0123ATGCATGC',
    'newline_inside': 'ATGCATGC\nATGCATGC',
}
# (seed, contig, L, invalid_ppm)
COUNT_SYNTH = {
    'synth_L64': (0, 0, 64, 0),
    'synth_L4999': (0, 1, 4999, 0),
    'synth_L5000': (0, 2, 5000, 0),
    'synth_L10000': (0, 3, 10000, 0),
    'synth_L5000_inv1pct': (1, 4, 5000, 10000),
    'synth_L70000': (2, 5, 70000, 0),          # longer than one wave pass
    'synth_L1000_inv20pct': (3, 6, 1000, 200000),
}
COUNT_KS = [1, 2, 3, 4, 5, 6]


def case_sequence(spec):
    if 'seq' in spec:
        return spec['seq']
    seed, c, L, ppm = spec['synth']
    return synth.synth_contig(seed, c, L, ppm)


def gen_counts(kmer, out):
    cases = {}
    for name, seq in COUNT_LITERALS.items():
        cases[name] = {'seq': seq}
    for name, spec in COUNT_SYNTH.items():
        cases[name] = {'synth': list(spec)}
    arrays = {}
    for name, spec in cases.items():
        seq = case_sequence(spec)
        for k in COUNT_KS:
            if len(seq) > 20000 and k not in (4, 5):
                continue
            arrays['%s__k%d' % (name, k)] = kmer.count_string(seq, k).astype(np.int64)
    # list form: kmer.count on a list (n > 1 -> 2-D; n == 1 -> 1-D)   scripts/kmer.py:93-105
    lst = [case_sequence(cases[n]) for n in ('mixed_invalid', 'len3', 'synth_L64', 'N_middle_runs', 'empty')]
    arrays['list5__k4'] = kmer.count(list(lst), 4).astype(np.int64)
    arrays['list1__k4'] = kmer.count([lst[0]], 4).astype(np.int64)
    # normalize=True through count_string (guarded) scripts/kmer.py:77-78
    arrays['norm_mixed_invalid__k4'] = kmer.count_string(lst[0], 4, normalize=True)
    arrays['norm_all_N__k4'] = kmer.count_string('N' * 40, 4, normalize=True)
    json.dump({'cases': cases, 'list5': ['mixed_invalid', 'len3', 'synth_L64', 'N_middle_runs', 'empty'],
               'ks': COUNT_KS}, open(os.path.join(out, 'count_cases.json'), 'w'), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(out, 'counts.npz'), **arrays)
    return len(arrays)


def gen_labels(kmer, out):
    doc = {
        'kmers': {str(k): kmer.kmers(k) for k in (1, 2, 3)},
        'kmers4_first8': kmer.kmers(4)[:8],
        'kmers4_last4': kmer.kmers(4)[-4:],
        'sequence_to_integers': {s: kmer.sequence_to_integers(s, 'ATGC')
                                 for s in ('ATGC', 'ATGCNatgc', 'GGCCTTAA-N', '')},
        'get_kmer_index': {s: kmer.get_kmer_index(s, 'ATGC') for s in ('AAAA', 'AAAT', 'CCCC', 'GATC', 'TGCA')},
    }
    json.dump(doc, open(os.path.join(out, 'labels.json'), 'w'), indent=1, sort_keys=True)


def gen_normalize(kmer, out):
    rng = np.random.default_rng(7)
    m = rng.integers(0, 5000, size=(12, 256)).astype(np.int64)
    m[3, :] = 0                       # zero row -> NaN (scripts/kmer.py:219-220)
    m[5, :] = 0
    m[5, 17] = 1
    m[7, :] = rng.integers(0, 400000, size=256)
    one_d = rng.integers(0, 99, size=64).astype(np.int64)
    with np.errstate(invalid='ignore', divide='ignore'):
        np.savez_compressed(os.path.join(out, 'normalize.npz'),
                            in2d=m, out2d=kmer.normalize_counts(m),
                            in1d=one_d, out1d=kmer.normalize_counts(one_d))


def gen_reference_features(fileIO, ref, out):
    d = os.path.join(ref, 'data', 'reference_features')
    pid, pc = fileIO.read_feature_file(os.path.join(d, 'positive_features.csv'))
    nid, nc = fileIO.read_feature_file(os.path.join(d, 'negative_features.csv'))
    assert pc.min() >= 0 and nc.min() >= 0 and max(pc.max(), nc.max()) < 2 ** 32
    np.savez_compressed(os.path.join(out, 'ref_features.npz'),
                        pos_ids=np.array(pid, dtype='U32'), pos_counts=pc.astype(np.uint32),
                        neg_ids=np.array(nid, dtype='U32'), neg_counts=nc.astype(np.uint32))
    return pc, nc


def gen_scoring(kmer, learning, phamer, pc, nc, out):
    """100 synthetic 5 kb contigs (seed 0) scored against the real reference matrix, through
    the reference's phamer.score_points (scripts/phamer.py:451-468)."""
    pos = kmer.normalize_counts(pc)
    neg = kmer.normalize_counts(nc)
    n_q = 100
    contigs = synth.synth_contigs(0, n_q, 5000)
    q_counts = kmer.count(list(contigs), 4)
    q = kmer.normalize_counts(q_counts)
    arrays = {'q_counts': q_counts.astype(np.int64), 'q': q}

    def run(tag, p, n):
        arrays['knn_' + tag] = phamer.score_points(q, p, n, method='knn')
        arrays['kmeans_' + tag] = phamer.score_points(q, p, n, method='kmeans')
        arrays['combo_' + tag] = phamer.score_points(q, p, n, method='combo')
        # the centroids score_points used (deterministic: KMeans(random_state=10))
        arrays['cpos_' + tag] = learning.get_centroids(p, learning.kmeans(p, 86))
        arrays['cneg_' + tag] = learning.get_centroids(n, learning.kmeans(n, 86))
        # neighbour indices / distances so near-ties are visible
        from sklearn.neighbors import NearestNeighbors
        train = np.vstack((p, n))
        dist, idx = NearestNeighbors(n_neighbors=6, algorithm='brute').fit(train).kneighbors(q)
        arrays['nbr_idx_' + tag] = idx.astype(np.int64)
        arrays['nbr_dist_' + tag] = dist

    # --equalize_reference (scripts/phamer.py:159-175): first min(n+, n-) rows of each
    sc = phamer.phamer_scorer()
    sc.positive_data, sc.negative_data = pos, neg
    sc.positive_ids, sc.negative_ids = np.arange(pos.shape[0]), np.arange(neg.shape[0])
    sc.equalize_reference_data()
    arrays['n_equalized'] = np.array([sc.positive_data.shape[0], sc.negative_data.shape[0]])
    run('eq', sc.positive_data, sc.negative_data)
    run('full', pos, neg)

    # queries that ARE reference rows / near-duplicates of them (distance-0 neighbours, ties)
    rng = np.random.default_rng(11)
    pick_p = rng.choice(pos.shape[0], 20, replace=False)
    pick_n = rng.choice(neg.shape[0], 20, replace=False)
    adv = np.vstack((pos[pick_p], neg[pick_n], 0.5 * (pos[pick_p[:10]] + neg[pick_n[:10]])))
    arrays['adv_q'] = adv
    arrays['adv_knn_full'] = phamer.score_points(adv, pos, neg, method='knn')
    arrays['adv_kmeans_full'] = phamer.score_points(adv, pos, neg, method='kmeans')
    arrays['adv_combo_full'] = phamer.score_points(adv, pos, neg, method='combo')

    # other neighbour counts through learning.knn (scripts/learning.py:118-128)
    train = np.vstack((pos, neg))
    labels = np.append(np.ones(pos.shape[0]), np.zeros(neg.shape[0]))
    for kn in (1, 5, 7):
        arrays['knn_full_kn%d' % kn] = learning.knn(q, train, labels, k=kn)
    np.savez_compressed(os.path.join(out, 'scoring_k4.npz'), **arrays)


def gen_scoring_highdim(kmer, learning, phamer, out):
    """k=5 (D=1024) and k=6 (D=4096) small cases: synthetic 'genomes' as the reference matrix
    (the shipped matrix is 4-mers only), half labelled positive."""
    arrays = {}
    for k, n_ref, n_q, L_ref in ((5, 240, 40, 20000), (6, 160, 24, 30000)):
        ref_counts = kmer.count(list(synth.synth_contigs(100 + k, n_ref, L_ref)), k)
        qc = kmer.count(list(synth.synth_contigs(200 + k, n_q, 10000)), k)
        # bias half of the reference rows so the two classes separate
        ref = kmer.normalize_counts(ref_counts)
        ref[: n_ref // 2, : ref.shape[1] // 4] *= 1.25
        ref = kmer.normalize_counts(ref)
        q = kmer.normalize_counts(qc)
        q[: n_q // 2, : q.shape[1] // 4] *= 1.25
        q = kmer.normalize_counts(q)
        p, n = ref[: n_ref // 2], ref[n_ref // 2:]
        tag = 'k%d' % k
        arrays['q_' + tag], arrays['pos_' + tag], arrays['neg_' + tag] = q, p, n
        arrays['knn_' + tag] = phamer.score_points(q, p, n, method='knn')
        sc = phamer.phamer_scorer()
        sc.k_clusters = 12
        sc.scoring_method = 'kmeans'
        sc.data_points, sc.positive_data, sc.negative_data = q, p, n
        arrays['kmeans_' + tag] = sc.score_points()
        arrays['cpos_' + tag] = learning.get_centroids(p, learning.kmeans(p, 12))
        arrays['cneg_' + tag] = learning.get_centroids(n, learning.kmeans(n, 12))
    np.savez_compressed(os.path.join(out, 'scoring_highdim.npz'), **arrays)


def gen_files(kmer, fileIO, out):
    """On-disk formats either side of the path, written by the reference's own writer functions
    (scripts/fileIO.py:169-181, 241-253; header text from scripts/basic.py:22-37).  The produced
    files are stored byte for byte (they are outputs, i.e. data)."""
    import tempfile
    g = np.load(os.path.join(out, 'scoring_k4.npz'))
    ids = np.array([str(i) for i in range(8)])
    counts = g['q_counts'][:8]
    scores = g['combo_eq'][:8]
    args = argparse.Namespace(input_file='contigs.fasta', kmer_length=4, output_file='contigs_4mers.csv',
                              symbols='ATGC', verbose=True, sample=None, file_identifier='.fna', debug=False)
    d = tempfile.mkdtemp()
    files = {}
    fileIO.save_counts(counts, ids, os.path.join(d, 'f_noargs.csv'))
    fileIO.save_counts(counts, ids, os.path.join(d, 'f_args.csv'), args=args)
    fileIO.save_phamer_scores(ids, scores, os.path.join(d, 's_noargs.csv'))
    fileIO.save_phamer_scores(ids, scores, os.path.join(d, 's_args.csv'), args=args)
    for name in ('f_noargs.csv', 'f_args.csv', 's_noargs.csv', 's_args.csv'):
        files[name] = open(os.path.join(d, name)).read()
    rid, rcounts = fileIO.read_feature_file(os.path.join(d, 'f_args.csv'))
    rid2, rnorm = fileIO.read_feature_file(os.path.join(d, 'f_args.csv'), normalize=True)
    sd = fileIO.read_phamer_output(os.path.join(d, 's_args.csv'))
    json.dump({'files': files, 'ids': ids.tolist(), 'scores_read_back': {k: sd[k] for k in sorted(sd)},
               'args': vars(args)}, open(os.path.join(out, 'files.json'), 'w'), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(out, 'files.npz'), counts=counts, scores=scores, read_counts=rcounts,
                        read_ids=np.array(rid, dtype='U16'), read_norm=rnorm)


def gen_transform(ref, out):
    """transform_kmers and its index tables (scripts/transform_kmers.py:21-88), executed with Python 2 integer
    division.  The tables are what the reference really computes -- they are NOT permutations (decompose() yields
    k+1 digits of which the first k are used, and the multipliers are k**i ascending): the fixture pins the
    drop-in to that behaviour, including the IndexError for k >= 5."""
    import math
    np_shim = types.ModuleType('np_shim')
    np_shim.__dict__.update(np.__dict__)
    np_shim.math = math           # the reference calls np.math.log (removed from NumPy 2)
    quiet = logging.getLogger('reference')
    tk = extract(os.path.join(ref, 'scripts', 'transform_kmers.py'),
                 ['DNA', 'get_transformed_indicies', 'get_reverse_complement_indicies', 'get_reverse_indicies',
                  'get_DNA_complement_indicies', 'transform_kmers'],
                 {'np': np_shim, 'xrange': range, 'logger': quiet}, py2_division=True)
    arrays = {}
    rng = np.random.default_rng(21)
    for k in (2, 3, 4):
        arrays['rev_idx_k%d' % k] = np.asarray(tk.get_reverse_indicies(k)).astype(np.int64)
        arrays['comp_idx_k%d' % k] = np.asarray(tk.get_DNA_complement_indicies(k)).astype(np.int64)
        arrays['revcomp_idx_k%d' % k] = np.asarray(tk.get_reverse_complement_indicies(k)).astype(np.int64)
        counts = rng.integers(0, 900, size=(6, 4 ** k)).astype(np.int64)
        arrays['in_k%d' % k] = counts
        arrays['rev_k%d' % k] = tk.transform_kmers(counts, reverse=True, complement=False)
        arrays['comp_k%d' % k] = tk.transform_kmers(counts, reverse=False, complement=True)
        arrays['revcomp_k%d' % k] = tk.transform_kmers(counts, reverse=True, complement=True)
        arrays['none_k%d' % k] = tk.transform_kmers(counts, reverse=False, complement=False)
    errors = {}
    for k in (5, 6):
        try:
            tk.transform_kmers(np.zeros((2, 4 ** k), dtype=np.int64), reverse=True, complement=True)
            errors[str(k)] = None
        except Exception as e:   # noqa: BLE001 -- the type is the fixture
            errors[str(k)] = type(e).__name__
    np.savez_compressed(os.path.join(out, 'transform.npz'), **arrays)
    json.dump({'errors_by_k': errors}, open(os.path.join(out, 'transform.json'), 'w'), indent=1, sort_keys=True)


ID_HEADERS = [
    'SuperContig_12_length_5000_ID_12', 'SuperContig_7_length_9000_ID_77-circular', '>SuperContig_3_ID_5',
    'x_ID_abc_ID_def', 'ID_9_ID_10', 'contig_ID_', 'a_ID_b-circular-circular', '  pad_ID_4  ',
    'gi|526245011|ref|NC_021865.1|', 'gi|1|gb|KC821634.1|', '>gi|2|ref|NC_000001.9|', 'a|b|c|d|', '||||',
    'NC_000913.3', 'CP009273.1', 'AE014075.1', 'NZ_CP011113.2', '>NC_1.1', 'X.1', '1.5', '12', 'abc', 'a.bc', '.1',
    'plain_contig_name', 'NC_000913.3|extra', 'a|b|c|d', 'a|b|c|d|e|f', 'scaffold_ID', 'IDX_ID_y_z', '',
]


def gen_ids(ref, out):
    """id_parser.get_id (scripts/id_parser.py:18-100) on the header shapes count_file meets: what comes out, or
    the exception type the reference raises."""
    quiet = logging.getLogger('reference')
    basic = extract(os.path.join(ref, 'scripts', 'basic.py'), ['represents_float'], {})
    idp = extract(os.path.join(ref, 'scripts', 'id_parser.py'),
                  ['get_contig_id', 'get_bacteria_id', 'get_phage_id', 'is_genbank_id', 'get_id'],
                  {'logger': quiet, 'basic': basic, 'os': os})
    cases = []
    for h in ID_HEADERS:
        try:
            cases.append({'header': h, 'id': idp.get_id(h), 'error': None})
        except Exception as e:   # noqa: BLE001
            cases.append({'header': h, 'id': None, 'error': type(e).__name__})
    json.dump({'cases': cases}, open(os.path.join(out, 'ids.json'), 'w'), indent=1, sort_keys=True)


def gen_cross_validation(ref, phamer, pc, nc, kmer, out):
    """cross_validator.cross_validate (scripts/cross_validate.py:57-101) with the reference's scoring function
    (phamer.score_points, as scripts/cross_validate.py:275 wires it), the real reference matrix equalised,
    N = 20, method 'combo'.  The reference shuffles with the global NumPy RNG, unseeded; the generator seeds it
    (np.random.seed) right before the call so that the fold assignment can be replayed by the test."""
    quiet = logging.getLogger('reference')
    cv = extract(os.path.join(ref, 'scripts', 'cross_validate.py'), ['cross_validator'],
                 {'np': np, 'xrange': range, 'logger': quiet})
    arrays = {}
    for tag, seed, N, method in (('n20_combo', 5, 20, 'combo'), ('n7_knn', 9, 7, 'knn')):
        v = cv.cross_validator()
        v.positive_data = kmer.normalize_counts(pc)
        v.negative_data = kmer.normalize_counts(nc)
        v.positive_ids = np.arange(pc.shape[0])
        v.negative_ids = np.arange(nc.shape[0])
        v.equalize_reference = True
        v.N = N
        v.method = method
        v.scoring_function = phamer.score_points
        np.random.seed(seed)
        ps, ns = v.cross_validate()
        # the fold assignment, replayed (same seed, same draw order as scripts/cross_validate.py:71-75)
        np.random.seed(seed)
        pa = np.arange(v.num_positive) % N
        na = np.arange(v.num_negative) % N
        np.random.shuffle(pa)
        np.random.shuffle(na)
        arrays['pos_scores_' + tag], arrays['neg_scores_' + tag] = ps, ns
        arrays['pos_asmt_' + tag], arrays['neg_asmt_' + tag] = pa.astype(np.int16), na.astype(np.int16)
        arrays['meta_' + tag] = np.array([seed, N, v.num_positive, v.num_negative])
    np.savez_compressed(os.path.join(out, 'cross_validation.npz'), **arrays)


def gen_distances(kmer, learning, pc, nc, out):
    """learning.distances / learning.closest_to (scripts/learning.py:47-66) on rows of the real matrix: 16 query rows
    (the scoring fixture's synthetic contigs and some reference rows themselves: zero distances, ties by duplicates)
    against 300 positive rows and 86 strided means."""
    with np.load(os.path.join(out, 'scoring_k4.npz')) as z:
        q = z['q'][:10]
    pos = kmer.normalize_counts(pc)[:300]
    neg = kmer.normalize_counts(nc)[:200]
    queries = np.vstack((q, pos[[0, 7, 299]], neg[[0, 5, 199]]))
    picks = np.stack([neg[i::43].mean(axis=0) for i in range(43)] + [neg[5], neg[5]])   # a duplicate row: a tie
    arrays = {'queries': queries, 'picks': picks,
              'dist_pos': np.stack([learning.distances(v, pos) for v in queries]),
              'dist_row_2d': learning.distances(queries[3:4], pos),
              'closest_picks': np.stack([learning.closest_to(v, picks) for v in queries]),
              'closest_idx': np.array([int(np.argmin(learning.distances(v, picks))) for v in queries])}
    np.savez_compressed(os.path.join(out, 'distances.npz'), **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(REPO, 'tests', 'golden'))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    kmer, learning, fileIO, phamer, skl = load_reference(args.ref)
    n = gen_counts(kmer, args.out)
    gen_labels(kmer, args.out)
    gen_normalize(kmer, args.out)
    pc, nc = gen_reference_features(fileIO, args.ref, args.out)
    gen_scoring(kmer, learning, phamer, pc, nc, args.out)
    gen_scoring_highdim(kmer, learning, phamer, args.out)
    gen_files(kmer, fileIO, args.out)
    gen_transform(args.ref, args.out)
    gen_ids(args.ref, args.out)
    gen_cross_validation(args.ref, phamer, pc, nc, kmer, args.out)
    gen_distances(kmer, learning, pc, nc, args.out)
    json.dump({'generator': 'tools/gen_golden.py', 'python': sys.version.split()[0],
               'numpy': np.__version__, 'scikit-learn': skl, 'count_arrays': n,
               'reference_functions': 'executed from the reference text via ast extraction; '
                                      'xrange=range shim; no reference text stored'},
              open(os.path.join(args.out, 'MANIFEST.json'), 'w'), indent=1, sort_keys=True)
    print("golden fixtures written to", args.out)


if __name__ == '__main__':
    main()
