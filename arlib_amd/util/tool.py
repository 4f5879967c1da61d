"""Host glue with the reference's names (util/tool.py): seeding contract, target selection, poison-data writer."""
import os
import random

import numpy as np
import torch


def seedSet(seed):
    """util/tool.py:101-108: seeds python `random` (the sampler's MT19937 stream), numpy and torch."""
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def targetItemSelect(data, arg, popularThreshold=0.1):
    """util/tool.py:52-99 (cache file under ./data/clean/<dataset>/ honoured when present)."""
    interact = data.matrix()
    itemNum = interact.shape[1]
    targetNum = int(arg.targetSize * itemNum) if arg.targetSize < 1 else int(arg.targetSize)
    path = './data/clean/' + str(data.dataName) + '/targetItem_' + arg.attackTargetChooseWay + '_' + str(targetNum) + '.txt'
    if os.path.exists(path):
        with open(path) as f:
            return [i.replace("'", '') for i in f.read().split(',')]
    order = np.argsort(np.asarray(interact.sum(0)))[0].tolist()
    if arg.attackTargetChooseWay == 'random':
        pool = list(range(itemNum))
    elif arg.attackTargetChooseWay == 'popular':
        pool = order[-int(popularThreshold * itemNum):]
    else:
        pool = order[:int(0.2 * itemNum)]
    # the reference samples from a *set* (util/tool.py:84-92); CPython 3.10 turns it into tuple(set) first, so the draw
    # depends on the set's iteration order -- reproduced literally
    targetItem = random.sample(tuple(set(pool)), targetNum)
    targetItem = [data.id2item[i] for i in targetItem]
    if os.path.isdir(os.path.dirname(path)):
        with open(path, 'w') as f:
            f.writelines(str(targetItem).replace('[', '').replace(']', ''))
    return targetItem


def dataSave(ratings, fileName, id2user, id2item, chunk=4_000_000):
    """util/tool.py:23-49: one `user item rating` line per non-zero interaction in `ratings.nonzero()` order, users missing from
    id2user written as fakeUser<row>, the value formatted as the matrix's scalar type formats itself (`1.0`, `0.5`).
    The reference builds the text with one sparse `ratings[i, j]` lookup per entry; here the triples come from the COO arrays
    and the lines are written chunk-wise through pandas' C writer (poisoned cfg2-size outputs have ~10^8 rows)."""
    import pandas as pd
    import scipy.sparse as sp
    m = sp.csr_matrix(ratings)
    m.sum_duplicates()
    coo = m.tocoo()
    keep = coo.data != 0
    rows, cols, data = coo.row[keep], coo.col[keep], coo.data[keep]
    n_u, n_i = m.shape
    users = np.array([id2user[i] if i in id2user else 'fakeUser' + str(i) for i in range(n_u)], dtype=object)
    items = np.array([id2item.get(j, '') for j in range(n_i)], dtype=object)
    if len(cols) and any(int(j) not in id2item for j in np.unique(cols)):
        raise KeyError('dataSave: an interacted item has no external id')
    uniq, inv = np.unique(data, return_inverse=True)
    vals = np.array(['{}'.format(v) for v in uniq], dtype=object)          # numpy scalar formatting, as '{}'.format(ratings[i, j])
    with open(fileName, 'w') as f:
        for lo in range(0, len(rows), chunk):
            hi = min(lo + chunk, len(rows))
            pd.DataFrame({'u': users[rows[lo:hi]], 'i': items[cols[lo:hi]], 'v': vals[inv[lo:hi]]}).to_csv(f, sep=' ', header=False, index=False, lineterminator='\n')
