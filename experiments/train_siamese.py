"""Siamese verification network trained with binary cross-entropy -- the MI355X counterpart of the reference's
experiments/train_siamese.py (filters 128, embedding 64, dropout 0, batch 64, Adam(clipnorm=1), 500-step epochs with
100 validation batches and 5-way 1-shot evaluation).     python -m experiments.train_siamese [--synthetic] ..."""
from experiments import _common as C
from voicemap_amd.keras_like import Adam
from voicemap_amd.models import build_siamese_net, get_baseline_convolutional_encoder
from voicemap_amd.utils import BatchPreProcessor, preprocess_instances


def main(argv=None):
    a = C.base_parser(__doc__).parse_args(argv)
    C.setup()
    train, valid = C.datasets(a, pad=a.pad)
    pre = BatchPreProcessor("siamese", preprocess_instances(a.downsampling))
    batches = lambda ds: (pre(b) for b in ds.yield_verification_batches(a.batchsize))
    train_batches = batches(train)
    workers = a.workers
    if a.device_data:  # same pairs, but the windows never exist on the host: offsets into an HBM-resident int16 buffer
        resident = C.device_resident(a, train)
        train_batches = (pre(b) for b in resident.yield_verification_batches_device(a.batchsize))
        workers = 0
    encoder = get_baseline_convolutional_encoder(a.filters, a.embedding_dimension, dropout=a.dropout, dtype=a.dtype)
    siamese = build_siamese_net(encoder, (C.input_length(a), 1), distance_metric="uniform_euclidean")
    siamese.compile(loss="binary_crossentropy", optimizer=Adam(clipnorm=1.), metrics=["accuracy"])
    siamese.summary()
    name = "siamese__filters_{}__embed_{}__drop_{}__pad={}".format(a.filters, a.embedding_dimension, a.dropout, a.pad)
    return siamese.fit_generator(generator=train_batches, steps_per_epoch=a.steps_per_epoch, validation_data=batches(valid),
                                 validation_steps=a.validation_steps, epochs=a.epochs, workers=workers,
                                 use_multiprocessing=True, callbacks=C.standard_callbacks(a, valid, pre, "siamese", name))


if __name__ == "__main__":
    main()
