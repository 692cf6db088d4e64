"""Siamese verification network trained with binary cross-entropy -- the MI355X counterpart of the reference's
experiments/train_siamese.py (filters 128, embedding 64, dropout 0, batch 64, Adam(clipnorm=1), 500-step epochs with
100 validation batches and 5-way 1-shot evaluation).     python -m experiments.train_siamese [--synthetic] ..."""
from experiments import _common as C
from voicemap_amd.keras_like import Adam
from voicemap_amd.models import build_siamese_net, get_baseline_convolutional_encoder, get_spectrogram_convolutional_encoder
from voicemap_amd.utils import BatchPreProcessor, preprocess_instances


def main(argv=None):
    ap = C.base_parser(__doc__)
    ap.add_argument("--frontend", default="waveform", choices=["waveform", "logmel"],
                    help="waveform: the reference's 1-D encoder on the decimated, whitened window; logmel: the log-mel + 2-D CNN variant "
                         "on the raw 16 kHz window (BASELINE.json config 4, not in the reference)")
    a = ap.parse_args(argv)
    if a.frontend == "logmel":
        a.downsampling = 1   # the front-end works on the raw window: no decimation, no whitening
    C.setup()
    train, valid = C.datasets(a, pad=a.pad)
    pre = BatchPreProcessor("siamese", preprocess_instances(a.downsampling, whitening=a.frontend != "logmel"))
    batches = lambda ds: (pre(b) for b in ds.yield_verification_batches(a.batchsize))
    train_batches = batches(train)
    workers = a.workers
    if a.device_data:  # same pairs, but the windows never exist on the host: offsets into an HBM-resident int16 buffer
        resident = C.device_resident(a, train)
        train_batches = (pre(b) for b in resident.yield_verification_batches_device(a.batchsize))
        workers = 0
    build = get_spectrogram_convolutional_encoder if a.frontend == "logmel" else get_baseline_convolutional_encoder
    encoder = build(a.filters, a.embedding_dimension, dropout=a.dropout, dtype=a.dtype)
    siamese = build_siamese_net(encoder, (C.input_length(a), 1), distance_metric="uniform_euclidean")
    siamese.compile(loss="binary_crossentropy", optimizer=Adam(clipnorm=1.), metrics=["accuracy"])
    siamese.summary()
    C.apply_sync_bn(a, siamese)
    name = "siamese__filters_{}__embed_{}__drop_{}__pad={}".format(a.filters, a.embedding_dimension, a.dropout, a.pad)
    if a.frontend == "logmel":
        name = "logmel_" + name
    return siamese.fit_generator(generator=train_batches, steps_per_epoch=a.steps_per_epoch, validation_data=batches(valid),
                                 validation_steps=a.validation_steps, epochs=a.epochs, workers=workers,
                                 use_multiprocessing=True, callbacks=C.standard_callbacks(a, valid, pre, "siamese", name))


if __name__ == "__main__":
    main()
