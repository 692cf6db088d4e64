"""Siamese network trained with the contrastive loss of Hadsell et al. -- counterpart of the reference's
experiments/siamese_contrastive_loss.py (filters 32, embedding 128, default dropout 0.05, batch 32, no ReduceLROnPlateau).
    python -m experiments.siamese_contrastive_loss [--synthetic] ..."""
from config import PATH
from experiments import _common as C
from voicemap_amd.keras_like import Adam, CSVLogger, ModelCheckpoint
from voicemap_amd.models import build_siamese_net, get_baseline_convolutional_encoder
from voicemap_amd.utils import BatchPreProcessor, NShotEvaluationCallback, contrastive_loss, preprocess_instances


def main(argv=None):
    p = C.base_parser(__doc__, batchsize=32, filters=32, embedding_dimension=128, dropout=0.05, epochs=25, pad=False)
    a = p.parse_args(argv)
    C.setup()
    train, valid = C.datasets(a, pad=False)
    whiten_downsample = BatchPreProcessor("siamese", preprocess_instances(a.downsampling, whitening=True))
    stream = lambda ds: (whiten_downsample(b) for b in ds.yield_verification_batches(a.batchsize))
    encoder = get_baseline_convolutional_encoder(a.filters, a.embedding_dimension, dropout=a.dropout, dtype=a.dtype)
    siamese = build_siamese_net(encoder, (C.input_length(a), 1))
    siamese.compile(loss=contrastive_loss, optimizer=Adam(clipnorm=1.), metrics=["accuracy"])
    key = "val_{}-shot_acc".format(a.n_shot)
    cbs = [NShotEvaluationCallback(a.num_evaluation_tasks, a.n_shot, a.k_way, valid, preprocessor=whiten_downsample),
           CSVLogger(PATH + "/logs/convnet_contrastive_loss.csv"),
           ModelCheckpoint(PATH + "/models/convnet_contrastive_loss.hdf5", monitor=key, mode="max", save_best_only=True, verbose=True)]
    return siamese.fit_generator(generator=stream(train), steps_per_epoch=a.steps_per_epoch, validation_data=stream(valid),
                                 validation_steps=a.validation_steps, epochs=a.epochs, workers=a.workers,
                                 use_multiprocessing=True, callbacks=cbs)


if __name__ == "__main__":
    main()
