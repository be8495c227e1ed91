from .nearest_neighbor import NearestNeighbor
