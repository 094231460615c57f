"""Mirror of the reference's pyramid builder (geotransformer/utils/data.py:13-77), device resident.

`precompute_data_stack_mode` makes the same 4 grid_subsample + 13 radius_search calls in the same
order with the same parameters; tensors stay on the GPU between calls (the reference runs this on
the CPU inside the DataLoader collate_fn and copies everything to the GPU afterwards)."""
import torch

from . import ext
from .ops import grid_subsample, radius_search




def precompute_data_stack_mode(points, lengths, num_stages, voxel_size, radius, neighbor_limits, order="reference",
                               pipeline=None, contiguous_neighbors=True):
    """utils/data.py:13-77: 4 grid_subsample + 13 radius_search calls, same order of results, same parameters.

    `pipeline=True`: the subsampling chain runs on a second host thread and a second HIP stream while this thread runs
    the radius searches of the levels that already exist.  That paid while every grid_subsample call ended in a host-side
    replay of libstdc++'s unordered_map iteration order (the GPU idled meanwhile); with the order evaluated on the device
    it is worth 4 % at 64 pairs per call on a quiet host and costs 3x when the two host threads fight for cores, so it is
    off by default.  Results are identical either way.
    `contiguous_neighbors=False`: index tensors may be column slices of wider rows (the reference's own truncated results
    are): no dense copy of a level's rows for a caller that does not need one."""
    assert num_stages == len(neighbor_limits)
    on_gpu = points.is_cuda
    if pipeline is None:
        pipeline = False
    points_list, lengths_list = [points], [lengths]
    neighbors_list, subsampling_list, upsampling_list = [], [], []

    def chain():
        p, l, v = points, lengths, voxel_size
        for i in range(1, num_stages):
            v *= 2  # data.py:22-28: level 0 is the input, level i uses voxel_size * 2^i
            p, l = grid_subsample(p, l, voxel_size=v, order=order)
            yield p, l

    if pipeline and on_gpu:
        import queue
        import threading
        dev = points.device
        main_stream = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        start = torch.cuda.Event()
        start.record(main_stream)
        q = queue.Queue()

        def worker():
            try:
                with torch.cuda.device(dev), torch.cuda.stream(side):
                    side.wait_event(start)  # the input cloud may still be in flight on the caller's stream
                    for p, l in chain():
                        ev = torch.cuda.Event()
                        ev.record(side)
                        q.put((p, l, ev))
            except BaseException as e:  # noqa: BLE001  (re-raised on the calling thread)
                q.put(e)

        th = threading.Thread(target=worker, name="gaussreg-pyramid-subsample", daemon=True)
        th.start()

        def next_level():
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            p, l, ev = item
            main_stream.wait_event(ev)
            p.record_stream(main_stream)  # allocated on the side stream, consumed here
            if l.is_cuda:
                l.record_stream(main_stream)
            points_list.append(p)
            lengths_list.append(l)
    else:
        gen = chain()

        def next_level():
            p, l = next(gen)
            points_list.append(p)
            lengths_list.append(l)
        th = None

    # radius search (data.py:31-69).  Level i's supports with radius r_i serve three searches (up-sampling of level
    # i-1, self, sub-sampling of level i+1): one SupportGrid per level bins them once.
    grids = [None] * num_stages

    def grid_of(i):
        if on_gpu and grids[i] is None:
            grids[i] = ext.SupportGrid(points_list[max(i - 1, 0)].shape[0])
        return grids[i]

    try:
        for i in range(num_stages):
            cur_points, cur_lengths = points_list[i], lengths_list[i]
            neighbors_list.append(radius_search(cur_points, cur_points, cur_lengths, cur_lengths, radius,
                                                neighbor_limits[i], grid=grid_of(i), contiguous=contiguous_neighbors))
            if i < num_stages - 1:
                next_level()
                sub_points, sub_lengths = points_list[i + 1], lengths_list[i + 1]
                subsampling_list.append(radius_search(sub_points, cur_points, sub_lengths, cur_lengths, radius,
                                                      neighbor_limits[i], grid=grid_of(i), contiguous=contiguous_neighbors))
                upsampling_list.append(radius_search(cur_points, sub_points, cur_lengths, sub_lengths, radius * 2,
                                                     neighbor_limits[i + 1], grid=grid_of(i + 1), contiguous=contiguous_neighbors))
            radius *= 2
    finally:
        if th is not None:
            th.join()
            # everything the side stream allocated (intermediates included) is settled before its thread-local scratch is
            # reused by anybody else; the worker's workspace entry is dropped with the thread
            side.synchronize()
            from . import _lib as _l
            for k in [k for k in _l._ws_cache if k[2] == side.cuda_stream]:
                del _l._ws_cache[k]
    return {'points': points_list, 'lengths': lengths_list, 'neighbors': neighbors_list,
            'subsampling': subsampling_list, 'upsampling': upsampling_list}


def _merge(data_dicts):
    import numpy as np
    collated = {}
    for d in data_dicts:
        for k, v in d.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            collated.setdefault(k, []).append(v)
    return collated


def single_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data=True,
                                 device=None):
    """data.py:80-137: one cloud per sample, [P_1 .. P_B] stacking.  `device`: where the pyramid is built and kept."""
    batch_size = len(data_dicts)
    collated = _merge(data_dicts)
    normals = torch.cat(collated.pop('normals'), dim=0) if 'normals' in collated else None
    feats = torch.cat(collated.pop('feats'), dim=0)
    points_list = collated.pop('points')
    lengths = torch.LongTensor([p.shape[0] for p in points_list])
    points = torch.cat(points_list, dim=0)
    if device is not None:
        points, feats = points.to(device), feats.to(device)
        normals = None if normals is None else normals.to(device)
    if batch_size == 1:
        for k, v in collated.items():
            collated[k] = v[0]
    if normals is not None:
        collated['normals'] = normals
    collated['features'] = feats
    if precompute_data:
        collated.update(precompute_data_stack_mode(points, lengths, num_stages, voxel_size, search_radius, neighbor_limits))
    else:
        collated['points'] = points
        collated['lengths'] = lengths
    collated['batch_size'] = batch_size
    return collated


def calibrate_neighbors_stack_mode(dataset, collate_fn, num_stages, voxel_size, search_radius, keep_ratio=0.8,
                                   sample_threshold=2000):
    """data.py:192-217: per-stage neighbour limits = the `keep_ratio` quantile of the neighbourhood sizes over (up to) one
    pass of the dataset.  The counting (`neighbors < neighbors.shape[0]`, exactly the reference's test, :207) and the
    histogram run on the device the collate function left the tensors on; only the (num_stages, hist_n) table is
    accumulated on the host."""
    import numpy as np
    hist_n = int(np.ceil(4 / 3 * np.pi * (search_radius / voxel_size + 1) ** 3))
    neighbor_hists = np.zeros((num_stages, hist_n), dtype=np.int32)
    max_neighbor_limits = [hist_n] * num_stages
    for i in range(len(dataset)):
        data_dict = collate_fn([dataset[i]], num_stages, voxel_size, search_radius, max_neighbor_limits, precompute_data=True)
        hists = []
        for neighbors in data_dict['neighbors']:
            counts = (neighbors < neighbors.shape[0]).sum(dim=1)
            hists.append(torch.bincount(counts, minlength=hist_n)[:hist_n].cpu().numpy())
        neighbor_hists += np.vstack(hists).astype(np.int32)
        if np.min(np.sum(neighbor_hists, axis=1)) > sample_threshold:
            break
    cum_sum = np.cumsum(neighbor_hists.T, axis=0)
    return np.sum(cum_sum < (keep_ratio * cum_sum[hist_n - 1, :]), axis=0)


def _reset_seed_worker_init_fn(worker_id):
    import random

    import numpy as np
    seed = torch.initial_seed() % (2 ** 32)  # utils/torch.py:39-44
    np.random.seed(seed)
    random.seed(seed)


def build_dataloader_stack_mode(dataset, collate_fn, num_stages, voxel_size, search_radius, neighbor_limits, batch_size=1,
                                num_workers=1, shuffle=False, drop_last=False, distributed=False, precompute_data=True):
    """data.py:219-249 (+ utils/torch.py:47-77).  With the pyramid built by HIP kernels the collate function needs the
    GPU: DataLoader workers are separate processes, so pass num_workers=0 (collate on the main process, tensors stay on
    the device) unless the collate function is given precompute_data=False."""
    from functools import partial
    sampler = torch.utils.data.DistributedSampler(dataset) if distributed else None
    return torch.utils.data.DataLoader(
        dataset, batch_size=batch_size, num_workers=num_workers, shuffle=False if distributed else shuffle, sampler=sampler,
        collate_fn=partial(collate_fn, num_stages=num_stages, voxel_size=voxel_size, search_radius=search_radius,
                           neighbor_limits=neighbor_limits, precompute_data=precompute_data),
        worker_init_fn=_reset_seed_worker_init_fn, pin_memory=False, drop_last=drop_last)


def registration_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                       precompute_data=True, device=None):
    """data.py:139-189 for the registration case: [ref_1..ref_B, src_1..src_B] stacking."""
    batch_size = len(data_dicts)
    collated = _merge(data_dicts)
    feats = torch.cat(collated.pop('ref_feats') + collated.pop('src_feats'), dim=0)
    points_list = collated.pop('ref_points') + collated.pop('src_points')
    lengths = torch.LongTensor([p.shape[0] for p in points_list])
    points = torch.cat(points_list, dim=0)
    if device is not None:
        points, feats = points.to(device), feats.to(device)
    if batch_size == 1:
        for k, v in collated.items():
            collated[k] = v[0]
    collated['features'] = feats
    if precompute_data:
        collated.update(precompute_data_stack_mode(points, lengths, num_stages, voxel_size, search_radius,
                                                   neighbor_limits))
    else:
        collated['points'] = points
        collated['lengths'] = lengths
    collated['batch_size'] = batch_size
    return collated
