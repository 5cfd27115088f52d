"""Training step of the ActivityNet1.3 recipe on MI355X (reference: AFSD/anet/train.py; BASELINE config 4).

`calc_bce_loss` / `forward_one_epoch` keep the reference's names and arithmetic (anet/train.py:136-186); the step
itself is thumos14.train.DetectorTrainer (flat arenas, bucketed RCCL all-reduce overlapped with backward, one-launch
Adam per optimizer group, optional HIP-graph replay) with the recipe's two optimizer groups: backbone at lr/10,
detection pyramid at lr (anet/train.py:304-312).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..thumos14.train import DetectorTrainer, total_cost  # noqa: F401


def calc_bce_loss(start, end, scores):
    """anet/train.py:136-144.  start/end (B,T,C) features; scores (B,3,T) = [action, start, end] masks."""
    start = torch.tanh(start).mean(-1)
    end = torch.tanh(end).mean(-1)
    loss_start = F.binary_cross_entropy(start.view(-1), scores[:, 1].contiguous().view(-1), reduction='mean')
    loss_end = F.binary_cross_entropy(end.view(-1), scores[:, 2].contiguous().view(-1), reduction='mean')
    return loss_start, loss_end


def forward_one_epoch(net, criterion, clips, targets, scores=None, training=True, ssl=True):
    """anet/train.py:147-186 with the criterion passed in (the reference uses a global CPD_Loss)."""
    if training:
        output_dict = net(clips, proposals=targets, ssl=ssl) if ssl else net(clips, ssl=False)
    else:
        with torch.no_grad():
            output_dict = net(clips)
    if ssl:
        anchor, positive, negative = output_dict
        weights = [1, 0.1, 0.1]
        loss_ = [nn.TripletMarginLoss()(anchor[i], positive[i], negative[i]) * weights[i] for i in range(3)]
        return torch.stack(loss_).sum(0)
    loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_act, loss_prop_act = criterion(
        [output_dict['loc'], output_dict['conf'], output_dict['prop_loc'], output_dict['prop_conf'],
         output_dict['center'], output_dict['priors'], output_dict['act'], output_dict['prop_act']], targets)
    src = getattr(output_dict, 'boundary_maps', None)
    if src is not None and src[0].is_cuda and src[0].dtype == torch.float32:
        # one launch per map: tanh, channel mean, BCE and the gradient, on the channel-major maps in place (csrc/bce.hip)
        from ..common.ops import BoundaryBCEFunction
        loss_start, loss_end = BoundaryBCEFunction.apply(src[0], scores, 1, 1)
        a, b = BoundaryBCEFunction.apply(src[1], scores, 1, 8)       # F.interpolate(scale_factor=1/8), nearest
        c, d = BoundaryBCEFunction.apply(src[2], scores, 1, 8)
    else:
        loss_start, loss_end = calc_bce_loss(output_dict['start'], output_dict['end'], scores)
        scores_ = scores[:, :, ::8]     # F.interpolate(scale_factor=1/8), nearest
        a, b = calc_bce_loss(output_dict['start_loc_prop'], output_dict['end_loc_prop'], scores_)
        c, d = calc_bce_loss(output_dict['start_conf_prop'], output_dict['end_conf_prop'], scores_)
    loss_start = loss_start + 0.1 * (a + c)
    loss_end = loss_end + 0.1 * (b + d)
    return loss_l, loss_c, loss_prop_l, loss_prop_c, loss_ct, loss_start, loss_end, loss_act, loss_prop_act


def optimizer_groups(net, learning_rate):
    """The two torch.optim.Adam groups of anet/train.py:304-312, in the reference's order."""
    return [(list(net.backbone.parameters()), learning_rate * 0.1),
            (list(net.coarse_pyramid_detection.parameters()), learning_rate)]


def make_trainer(net, criterion, loss_weights, learning_rate=1e-4, weight_decay=1e-4, **kw):
    return DetectorTrainer(net, criterion, loss_weights, learning_rate, weight_decay,
                           param_groups=optimizer_groups(net, learning_rate), forward_fn=forward_one_epoch, **kw)


# ----------------------------------------------------------------------------- the training driver (anet/train.py:282-350)
def build_training(config, device, random_init=False, dist_group=None, **trainer_kw):
    """Model, criterion and trainer from a parsed config: BDNet + MultiSegmentLoss(num_cls, piou, 1.0, cls_loss_type,
    edl_config, os_head) + Adam with the backbone at lr / 10 (anet/train.py:287-318)."""
    from .BDNet import BDNet, model_cfg_from
    from .multisegment_loss import MultiSegmentLoss
    tr, md = config['training'], config['model']
    net = BDNet(in_channels=md['in_channels'], backbone_model=md.get('backbone_model'), training=not random_init,
                frame_num=config['dataset']['training']['clip_length'], use_edl=md.get('use_edl', False), cfg=model_cfg_from(config))
    if random_init:                 # no pretrained I3D file (synthetic runs): the architecture's glorot initialisation
        net.backbone._model.apply(BDNet.weight_init)
    net = net.to(device).train()
    os_head = md.get('os_head', False)
    num_cls = config['dataset']['num_classes'] - 1 if os_head else config['dataset']['num_classes']
    cls_loss_type = 'edl' if tr.get('edl_loss', False) else 'focal'      # anet/train.py:24 (no overwrite in this recipe)
    crit = MultiSegmentLoss(num_cls, tr['piou'], 1.0, cls_loss_type=cls_loss_type, edl_config=tr.get('edl_config'),
                            os_head=os_head).to(device)
    weights = dict(lw=tr['lw'], cw=tr['cw'], ctw=tr['ctw'], actw=tr.get('actw', 1.0), ssl=tr['ssl'])
    trainer = make_trainer(net, crit, weights, learning_rate=tr['learning_rate'], weight_decay=tr['weight_decay'],
                           process_group=dist_group, **trainer_kw)
    return net, crit, trainer


def main(argv=None):
    """python -m opental_amd.anet.train configs/anet_opental.yaml --open_set --split 0 --lw 1 --cw 1 --piou 0.6 [--resume N]

    The reference's command line (AFSD/anet/README.md:61; AFSD/common/config.py:10-37) plus --random_init, --save_after N,
    --max_steps N, --launch lanes|eager as in opental_amd.thumos14.train.  One process per GPU; under torchrun the ranks all-reduce gradients
    over RCCL (DetectorTrainer).  The epoch loop is thumos14.train.run_one_epoch: same batches-by-permutation sampler,
    ssl branch when the first sample's splice succeeded (`if flags[0]`, anet/train.py:223), device-side loss sums."""
    import os
    import sys
    import torch.distributed as dist
    from ..common import anet_dataset as D
    from ..common import config as C
    from ..common import ops
    from ..common.thumos_dataset import ClipStager, max_target_count
    from ..thumos14.train import run_one_epoch, set_seed
    argv = list(sys.argv[1:] if argv is None else argv)
    extra = {'random_init': False, 'save_after': 10, 'max_steps': None, 'launch': 'lanes'}
    rest, i = [], 0
    while i < len(argv):
        a = argv[i]
        if a == '--random_init':
            extra['random_init'] = True
        elif a in ('--save_after', '--max_steps'):
            extra[a[2:]] = int(argv[i + 1]); i += 1
        elif a == '--launch':
            extra['launch'] = argv[i + 1]; i += 1
        else:
            rest.append(a)
        i += 1
    config = C.set_config(C.get_config(rest))
    tr = config['training']
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit("opental_amd.anet.train needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', device_id=dev)
    ops.CONV_PRECISION = 1 if os.environ.get('OTAL_DTYPE', 'bf16') == 'bf16' else 0
    set_seed(tr['random_seed'])
    net, crit, trainer = build_training(config, dev, extra['random_init'])
    ds = config['dataset']['training']
    dataset = D.ANET_Dataset(ds['video_info_path'], ds['video_mp4_path'], ds['clip_length'], ds['crop_size'], ds['clip_stride'],
                             channels=config['model']['in_channels'], binary_class=config['dataset']['num_classes'] == 2)
    if len(dataset) == 0:
        raise SystemExit("no training videos found under " + str(ds['video_mp4_path']))
    any_shape = dataset.video_shape(dataset.training_list[0]['video_name'])
    stager = ClipStager(tr['batch_size'], ds['clip_length'], int(any_shape[1]), int(any_shape[2]), ds['crop_size'],
                        device=dev, max_targets=max_target_count(dataset), score_rows=3,
                        copy_stream=ops.side_wgrads(dev).side)
    if extra['launch'] not in ('lanes', 'eager'):
        raise SystemExit("--launch takes lanes or eager")
    trainer.launch = extra['launch']
    checkpoint_path = tr['checkpoint_path']
    train_state_path = os.path.join(checkpoint_path, 'training')
    start_epoch = trainer.resume_training(tr['resume'], checkpoint_path, train_state_path)
    if rank == 0:
        print(f"batch size: {tr['batch_size']}  learning rate: {tr['learning_rate']} (backbone x0.1)  weight decay: {tr['weight_decay']}  "
              f"max epoch: {tr['max_epoch']}  cls loss: {crit.cls_loss_type}  clips: {len(dataset)}  ranks: {world}  resume: {tr['resume']}")
    history = []
    for epoch in range(start_epoch, tr['max_epoch'] + 1):
        if crit.cls_loss_type == 'edl':
            crit.cls_loss.epoch = epoch
            crit.cls_loss.total_epoch = tr['max_epoch']
        v = run_one_epoch(epoch, trainer, dataset, stager, tr['batch_size'], rank, world, extra['max_steps'],
                          log=print if rank == 0 else (lambda *a: None))
        history.append(v)
        if epoch > extra['save_after'] and rank == 0:
            trainer.save_model(epoch, checkpoint_path, train_state_path)
    if world > 1:
        dist.barrier()
    return trainer, history


if __name__ == '__main__':
    main()
