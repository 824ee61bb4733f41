"""Drop-in for the reference's top-level 2D package (``retinanet/`` = upstream yhenon/pytorch-retinanet):
put ``3d-playground_amd/flat2d`` AND ``3d-playground_amd`` on ``sys.path`` where a script put the reference root."""
