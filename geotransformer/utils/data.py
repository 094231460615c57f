from gaussreg_amd.data import precompute_data_stack_mode, registration_collate_fn_stack_mode  # noqa: F401
