"""``NearestNeighbors`` placeholder; implemented in knn.py once the search kernels land."""


class NearestNeighbors():
    def __init__(self, *a, **k):
        raise NotImplementedError("NearestNeighbors: HIP k-NN path not built yet")
