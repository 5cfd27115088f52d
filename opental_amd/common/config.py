"""Config loader with the reference's YAML schema and command-line flags
(AFSD/common/config.py:5-98), loaded by an explicit call instead of at import time.

    cfg = get_config(['configs/thumos14_opental_final.yaml', '--open_set', '--split', '0'])

For drop-in use the module attribute ``config`` still exists: it is resolved lazily on first
access (from sys.argv, exactly like the reference's import-time ``config = get_config()``,
config.py:101) or after ``set_config(cfg)``.
"""
import argparse
import sys

import yaml

_FLAGS = [  # (flag, kwargs)   -- the reference's CLI surface, config.py:10-37
    ('--batch_size', dict(type=int)), ('--learning_rate', dict(type=float)),
    ('--weight_decay', dict(type=float)), ('--max_epoch', dict(type=int)),
    ('--checkpoint_path', dict(type=str)), ('--seed', dict(type=int)),
    # the reference declares type=bool, so any non-empty string is True;
    # parsed the same way on purpose
    ('--focal_loss', dict(type=bool)),
    ('--nms_thresh', dict(type=float)), ('--nms_sigma', dict(type=float)), ('--top_k', dict(type=int)),
    ('--output_json', dict(type=str)),
    ('--lw', dict(type=float, default=1.0)), ('--cw', dict(type=float, default=10.0)),
    ('--ctw', dict(type=float, default=1.0)), ('--actw', dict(type=float, default=1.0)),
    ('--ssl', dict(type=float, default=0.1)), ('--piou', dict(type=float, default=0)),
    ('--resume', dict(type=int, default=0)), ('--ngpu', dict(type=int, default=1)),
    ('--fusion', dict(action='store_true')), ('--open_set', dict(action='store_true')),
    ('--split', dict(type=int, choices=[0, 1, 2, 3, 4], default=0)),
    ('--ood_scoring', dict(type=str, default='confidence',
                           choices=['uncertainty', 'confidence', 'uncertainty_actionness', 'a_by_inv_u',
                                    'u_by_inv_a', 'half_au'])),
    ('--exp_tag', dict(type=str, default=None)),
]
_TRAIN_OVERRIDES = ('batch_size', 'learning_rate', 'weight_decay', 'max_epoch')
_TEST_OVERRIDES = ('nms_thresh', 'nms_sigma', 'top_k', 'output_json', 'exp_tag')
_SPLIT_PATHS = (('dataset', 'class_info_path'), ('dataset', 'training', 'video_anno_path'),
                ('dataset', 'testing', 'video_anno_path'), ('training', 'checkpoint_path'),
                ('testing', 'checkpoint_path'), ('testing', 'output_path'))
_SPLIT_PATHS_IF_TAGGED = (('dataset', 'training', 'video_info_path'), ('dataset', 'testing', 'video_info_path'))

_config = None


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('config_file', type=str, default='configs/default.yaml', nargs='?')
    for flag, kw in _FLAGS:
        p.add_argument(flag, **kw)
    return p


def _get(d, path):
    for k in path[:-1]:
        d = d[k]
    return d, path[-1]


def get_config(argv=None):
    args = build_parser().parse_args(argv)
    with open(args.config_file, 'r', encoding='utf-8') as f:
        data = yaml.load(f.read(), Loader=yaml.FullLoader)
    tr, te = data['training'], data['testing']
    tr['learning_rate'] = float(tr['learning_rate'])
    tr['weight_decay'] = float(tr['weight_decay'])
    for k in _TRAIN_OVERRIDES:
        v = getattr(args, k)
        if v is not None:
            tr[k] = type(tr.get(k, v))(v) if k in tr else v
    if args.checkpoint_path is not None:
        tr['checkpoint_path'] = te['checkpoint_path'] = args.checkpoint_path
    if args.seed is not None:
        tr['random_seed'] = args.seed
    if args.focal_loss is not None:
        tr['focal_loss'] = args.focal_loss
    for k in ('lw', 'cw', 'ctw', 'actw', 'ssl', 'piou', 'resume'):
        tr[k] = getattr(args, k)
    data['ngpu'] = args.ngpu
    te['fusion'], te['split'], te['ood_scoring'] = args.fusion, args.split, args.ood_scoring
    for k in _TEST_OVERRIDES:
        v = getattr(args, k)
        if v is not None:
            te[k] = v
    data['open_set'] = args.open_set
    if args.open_set:   # '{id:d}' placeholders take the split id (config.py:83-96)
        for path in _SPLIT_PATHS:
            d, k = _get(data, path)
            d[k] = d[k].format(id=args.split)
        for path in _SPLIT_PATHS_IF_TAGGED:
            d, k = _get(data, path)
            if 'split_' in d[k]:
                d[k] = d[k].format(id=args.split)
    return data


def set_config(cfg):
    global _config
    _config = cfg
    return cfg


def __getattr__(name):   # PEP 562: `from opental_amd.common.config import config`
    global _config
    if name == 'config':
        if _config is None:
            _config = get_config(sys.argv[1:])
        return _config
    raise AttributeError(name)
