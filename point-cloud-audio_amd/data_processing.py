"""ESC-50 metadata helpers: drop-in for ``Code/data_processing.py`` (``load_esc``, ``tt_split``).

File-system metadata only (a CSV and a list of paths) - not on the accelerated path; kept so
that ``from data_processing import *`` in the reference's train / eval scripts
(Code/settransformer.py:25, Code/pceval.py:16, Code/baseline_eval.py:16) resolves here.  The
split consumes the global numpy RNG exactly as the reference does (one
``np.random.permutation`` per class, classes in ascending order), so a script that seeds numpy
gets the reference's train / test partition.
"""
import numpy as np

__all__ = ["load_esc", "tt_split", "ESC10_CATEGORIES"]

ESC10_CATEGORIES = ['dog', 'chainsaw', 'crackling_fire', 'helicopter', 'rain', 'crying_baby',
                    'clock_tick', 'sneezing', 'rooster', 'sea_waves']


def load_esc(loc='../ESC-50-master/meta/esc50.csv', loc_audio='../ESC-50-master/audio/',
             list_categories=ESC10_CATEGORIES):
    """Rows of ``esc50.csv`` whose ``category`` is in ``list_categories`` (default: the ESC-10
    subset) -> (array of ``loc_audio + filename``, int array of labels), a label being the
    category's position in ``list_categories``.  Row order of the CSV is kept.
    (Code/data_processing.py:8-38)"""
    import pandas as pd
    meta = pd.read_csv(loc)
    meta = meta[meta.category.isin(list_categories)]
    index_of = {c: i for i, c in enumerate(list_categories)}
    labels = np.array([index_of[c] for c in meta.category.to_list()])
    paths = np.array([loc_audio + f for f in meta.filename.to_list()])
    return paths, labels


def tt_split(list_audio_locs, l, f=0.8):
    """Per-class random split BY FILE (not by frame): for each class in ascending order a
    ``np.random.permutation`` of its files, the first ``int(f * n)`` go to train, the rest to
    test.  Returns (audio_train, l_train, audio_test, l_test) as lists.
    (Code/data_processing.py:40-65)"""
    nclass = max(l) + 1
    per_class = [[] for _ in range(nclass)]
    for path, lab in zip(list_audio_locs, l):
        per_class[lab].append(path)
    audio_train, l_train, audio_test, l_test = [], [], [], []
    for k, files in enumerate(per_class):
        order = np.random.permutation(len(files))
        cut = int(f * len(files))
        audio_train += [files[i] for i in order[:cut]]
        l_train += [k] * cut
        audio_test += [files[i] for i in order[cut:]]
        l_test += [k] * (len(files) - cut)
    return audio_train, l_train, audio_test, l_test
