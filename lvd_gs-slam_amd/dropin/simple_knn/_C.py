from lvdgs.simple_knn import distCUDA2  # noqa: F401
