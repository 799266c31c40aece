from .prepare import ProcessKITTIODOM, ProcessRELLIS, preproc_gt, preproc_img, preproc_pcd, rand_init_params  # noqa: F401
