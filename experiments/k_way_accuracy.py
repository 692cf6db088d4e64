"""k-way n-shot accuracy sweep of trained models -- counterpart of the reference's experiments/k_way_accuracy.py
(k = 2..20, n in {1, 5}, 1000 tasks each, distance 'dot_product'); results are appended to a CSV as they arrive and the finished table replaces
them at the end, as in the reference (:46-72).
    python -m experiments.k_way_accuracy --siamese models/x.hdf5 [--classifier models/y.hdf5] [--synthetic]"""
import argparse

import pandas as pd

from config import PATH
from voicemap_amd.librispeech import LibriSpeechDataset, SyntheticSpeechDataset
from voicemap_amd.models import load_model
from voicemap_amd.utils import BatchPreProcessor, n_shot_task_evaluation, preprocess_instances


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__)
    p.add_argument("--siamese", required=True)
    p.add_argument("--classifier", default=None)
    p.add_argument("--downsampling", type=int, default=4)
    p.add_argument("--n-seconds", type=float, default=3)
    p.add_argument("--validation-set", default="dev-clean")
    p.add_argument("--k-way", type=int, nargs="+", default=list(range(2, 21)))
    p.add_argument("--n-shot", type=int, nargs="+", default=[1, 5])
    p.add_argument("--num-tasks", type=int, default=1000)
    p.add_argument("--distance", default="dot_product", choices=["euclidean", "cosine", "dot_product"])
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--cached", action="store_true",
                   help="embed the evaluation set ONCE per model and run every (k, n) cell on the cached (N, E) matrix "
                        "(voicemap_amd/retrieval.py: tasks are row indices, one launch per cell; each window whitened alone)")
    p.add_argument("--device-sampler", action="store_true", help="with --cached: draw the tasks on the GPU (same distribution, not "
                                                                 "the reference's np.random sequence)")
    a = p.parse_args(argv)
    # under torchrun the tasks of every (k, n) cell are sharded over the ranks (BASELINE.json config 5); rank 0 writes the CSV
    from experiments._common import setup
    rank, _ = setup()
    if a.synthetic:
        valid = SyntheticSpeechDataset(num_speakers=40, files_per_speaker=12, seconds=a.n_seconds, stochastic=False, seed=1)
    else:
        valid = LibriSpeechDataset(a.validation_set, a.n_seconds, stochastic=False)
    pre = BatchPreProcessor("siamese", preprocess_instances(a.downsampling))
    nets = [("siamese", "siamese", load_model(a.siamese))]
    if a.classifier:
        nets.append(("classifier", "classifier", load_model(a.classifier)))
    out = PATH + "/logs/k-way_n-shot_accuracy_{}_{}.csv".format(a.validation_set, a.distance)
    rows = []
    caches, sampler = {}, None
    if a.cached:
        from voicemap_amd import retrieval
        for method, kind, net in nets:
            caches[method] = retrieval.embed_corpus(net, valid, pre, kind)      # sharded over ranks + all-gathered under torchrun
        if a.device_sampler:
            sampler = retrieval.DeviceTaskSampler(valid, caches["siamese"].emb.device, seed=rank)
    if rank == 0:
        with open(out, "w") as f:
            f.write("method,n_correct,n_tasks,n_shot,k_way\n")
    for k in a.k_way:
        for n in a.n_shot:
            for method, kind, net in nets:
                if a.cached:
                    correct = retrieval.n_shot_task_evaluation_cached(net, valid, pre, a.num_tasks, n, k, kind, a.distance,
                                                                      cache=caches[method], sampler=sampler or "reference")
                else:
                    correct = n_shot_task_evaluation(net, valid, pre, a.num_tasks, n, k, network_type=kind, distance=a.distance)
                # (the reference's table calls the classifier's rows 'bottleneck', its intermediate lines 'classifier': k_way_accuracy.py:64-69)
                rows.append({"method": "bottleneck" if method == "classifier" else method, "n_correct": correct, "n_tasks": a.num_tasks,
                             "n": n, "k": k})
                if rank == 0:
                    with open(out, "a") as f:
                        f.write("{},{},{},{},{}\n".format(method, correct, a.num_tasks, n, k))
    results = pd.DataFrame(rows)
    if rank == 0:
        results.to_csv(out, index=False)     # like the reference (:71-72): the finished table replaces the intermediate lines
    return results


if __name__ == "__main__":
    main()
