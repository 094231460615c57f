"""Mirror of the reference's pyramid builder (geotransformer/utils/data.py:13-77), device resident.

`precompute_data_stack_mode` makes the same 4 grid_subsample + 13 radius_search calls in the same
order with the same parameters; tensors stay on the GPU between calls (the reference runs this on
the CPU inside the DataLoader collate_fn and copies everything to the GPU afterwards)."""
import torch

from . import ext
from .ops import grid_subsample, radius_search


def precompute_data_stack_mode(points, lengths, num_stages, voxel_size, radius, neighbor_limits, order="reference"):
    assert num_stages == len(neighbor_limits)
    points_list, lengths_list = [], []
    neighbors_list, subsampling_list, upsampling_list = [], [], []
    # grid subsampling (data.py:22-28: voxel doubles every stage, level 0 is the input)
    for i in range(num_stages):
        if i > 0:
            points, lengths = grid_subsample(points, lengths, voxel_size=voxel_size, order=order)
        points_list.append(points)
        lengths_list.append(lengths)
        voxel_size *= 2
    # radius search (data.py:31-69).  Level i's supports with radius r_i serve three searches (up-sampling of level
    # i-1, self, sub-sampling of level i+1): one SupportGrid per level bins them once.
    on_gpu = points_list[0].is_cuda
    grids = [ext.SupportGrid(points_list[max(i - 1, 0)].shape[0]) if on_gpu else None for i in range(num_stages)]
    for i in range(num_stages):
        cur_points, cur_lengths = points_list[i], lengths_list[i]
        neighbors_list.append(radius_search(cur_points, cur_points, cur_lengths, cur_lengths, radius,
                                            neighbor_limits[i], grid=grids[i]))
        if i < num_stages - 1:
            sub_points, sub_lengths = points_list[i + 1], lengths_list[i + 1]
            subsampling_list.append(radius_search(sub_points, cur_points, sub_lengths, cur_lengths, radius,
                                                  neighbor_limits[i], grid=grids[i]))
            upsampling_list.append(radius_search(cur_points, sub_points, cur_lengths, sub_lengths, radius * 2,
                                                 neighbor_limits[i + 1], grid=grids[i + 1]))
        radius *= 2
    return {'points': points_list, 'lengths': lengths_list, 'neighbors': neighbors_list,
            'subsampling': subsampling_list, 'upsampling': upsampling_list}


def registration_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                       precompute_data=True, device=None):
    """data.py:139-189 for the registration case: [ref_1..ref_B, src_1..src_B] stacking."""
    import numpy as np
    batch_size = len(data_dicts)
    collated = {}
    for d in data_dicts:
        for k, v in d.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            collated.setdefault(k, []).append(v)
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
