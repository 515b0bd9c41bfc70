"""MI355X-native drop-in for the reference's `simple_knn` package (un-vendored submodule submodules/simple-knn,
.gitmodules:1-3).  The only symbol the reference uses is `simple_knn._C.distCUDA2`
(scene/gaussian_model.py:21,159); see `_C.py`."""
