"""Command-line front end shared by the plugin mirrors (Serra09, EarlySNF, FTM2D): the option names of the reference's
scripts (benchmarking/Serra09.py:218-244 and friends), so existing job scripts keep working."""
import argparse


def run(make_algorithm, description, chroma_default, shortname_default, batch_options=False):
    """make_algorithm(args, do_memmaps) -> CoverAlgorithm instance."""
    ap = argparse.ArgumentParser(description=description + " (MI355X path)",
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("-d", "--datapath", default="../features_covers80", help="directory with one feature file per song")
    ap.add_argument("-s", "--shortname", default=shortname_default, help="dataset tag used in cache and result file names")
    ap.add_argument("-c", "--chroma_type", default=chroma_default, help="which chroma field of the feature files to use")
    ap.add_argument("-p", "--parallel", type=int, choices=(0, 1), default=0, help="ignored: the GPU batch is the parallelism")
    ap.add_argument("-n", "--n_cores", type=int, default=1, help="ignored")
    if batch_options:
        ap.add_argument("-r", "--range", default="", help="'<blocks>-<index>': compute one block of the pair grid only")
        ap.add_argument("-f", "--features", type=int, choices=(0, 1), default=0, help="with --range: only cache that slice's features")
        ap.add_argument("-w", "--wsub", type=int, default=-1, help="with --range: checkpoint in sub-blocks of this width")
        ap.add_argument("-b", "--batch_path", default="", help="evaluate from the block files with this prefix")
    args = ap.parse_args()
    rng = getattr(args, "range", "")
    alg = make_algorithm(args, len(rng) == 0)
    if getattr(args, "batch_path", ""):
        alg.load_batches(args.batch_path)
    elif rng:
        blocks, index = (int(v) for v in rng.split("-"))
        if args.features == 1:
            alg.do_batch_features(blocks, index)
        else:
            alg.do_batch(blocks, index, args.wsub)
        print("... Done ....")
        return alg
    else:
        alg.all_pairwise(args.parallel, args.n_cores, symmetric=True)
    for kind in alg.Ds.keys():
        print(kind)
        alg.getEvalStatistics(kind)
    if not getattr(args, "batch_path", ""):
        alg.cleanup_memmap()
    print("... Done ....")
    return alg
