# Same three bindings as the reference package (PatchPerPix/vote_instances/__init__.py:1-3):
# the second import of `main` wins, so `vote_instances.main` is the blockwise driver's main.
from .vote_instances import main
from .stitch_patch_graph import main, get_offsets, get_offset_str, load_input, verify_shape, write_output, write_output, clean_mask
from .graph_to_labeling import affGraphToInstances
