"""Speaker-identification classifier on the same encoder (+ Dense(num_classes, softmax), categorical cross-entropy) --
counterpart of the reference's experiments/train_classifier.py; its bottleneck layer is evaluated with the same
n-shot tasks (mode='classifier').     python -m experiments.train_classifier [--synthetic] ..."""

import numpy as np

from experiments import _common as C
from voicemap_amd.keras_like import Adam, Dense, Sequence, to_categorical
from voicemap_amd.models import get_baseline_convolutional_encoder
from voicemap_amd.utils import BatchPreProcessor, preprocess_instances


class ShuffledBatches(Sequence):
    """Batches of (windows, labels) over a dataset in a permutation that is redrawn every epoch."""

    def __init__(self, dataset, preprocessor, batchsize):
        self.dataset, self.preprocessor, self.batchsize = dataset, preprocessor, batchsize
        self.order = np.random.permutation(len(dataset))

    def __len__(self):
        return len(self.dataset) // self.batchsize

    def __getitem__(self, item):
        picks = [self.dataset[i] for i in self.order[item * self.batchsize:(item + 1) * self.batchsize]]
        windows = np.stack([w[:, np.newaxis] for w, _ in picks])
        labels = np.stack([l for _, l in picks])[:, np.newaxis]
        return self.preprocessor((windows, labels))

    def on_epoch_end(self):
        self.order = np.random.permutation(len(self.dataset))


def main(argv=None):
    a = C.base_parser(__doc__, pad=False, dropout=None).parse_args(argv)
    C.setup()
    train, valid = C.datasets(a, pad=False)
    ids = sorted(train.df["speaker_id"].unique())
    index_of = {s: i for i, s in enumerate(ids)}
    one_hot = lambda y: to_categorical(np.array([index_of[s] for s in y[:, 0]]), train.num_classes())
    pre = BatchPreProcessor("classifier", preprocess_instances(a.downsampling), one_hot)
    # the reference defines dropout = 0.0 (train_classifier.py:28) but never passes it to the build function (:110), so its
    # classifier trains at the function's default SpatialDropout1D rate 0.05 while the run is NAMED drop_0.0 (:40).  Same here
    # unless --dropout is given explicitly.
    rate = 0.05 if a.dropout is None else a.dropout   # default None: "--dropout=0.1" and argparse abbreviations count as explicit
    a.dropout = 0.0 if a.dropout is None else a.dropout   # the name the reference gives the run (:28,:40)
    classifier = get_baseline_convolutional_encoder(a.filters, a.embedding_dimension, (C.input_length(a), 1), dropout=rate,
                                                    dtype=a.dtype)
    classifier.add(Dense(train.num_classes(), activation="softmax"))
    classifier.compile(loss="categorical_crossentropy", optimizer=Adam(clipnorm=1.), metrics=["accuracy"])
    classifier.summary()
    C.apply_sync_bn(a, classifier)
    name = "classifier__filters_{}__embed_{}__drop_{}__pad={}".format(a.filters, a.embedding_dimension, a.dropout, a.pad)
    # twice the siamese step count: a siamese batch carries two windows per sample (reference comment, :126-127)
    return classifier.fit_generator(generator=ShuffledBatches(train, pre, a.batchsize), steps_per_epoch=2 * a.steps_per_epoch,
                                    epochs=a.epochs, workers=a.workers, use_multiprocessing=True,
                                    callbacks=C.standard_callbacks(a, valid, pre, "classifier", name, plateau_patience=5))


if __name__ == "__main__":
    main()
