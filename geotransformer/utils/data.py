from gaussreg_amd.data import (  # noqa: F401
    build_dataloader_stack_mode,
    calibrate_neighbors_stack_mode,
    precompute_data_stack_mode,
    registration_collate_fn_stack_mode,
    single_collate_fn_stack_mode,
)
