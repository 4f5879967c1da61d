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


def dataSave(ratings, fileName, id2user, id2item):
    """util/tool.py:23-49: one `user item rating` line per stored interaction, CSR order."""
    coo = ratings.tocoo() if hasattr(ratings, 'tocoo') else None
    rows, cols = ratings.nonzero()
    with open(fileName, 'w') as f:
        for i, j in zip(rows.tolist(), cols.tolist()):
            user = id2user[i] if i in id2user else 'fakeUser' + str(i)
            f.write('{} {} {}\n'.format(user, id2item[j], ratings[i, j]))
