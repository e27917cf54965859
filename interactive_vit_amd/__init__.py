"""MI355X-native ViT forward engine behind the interactive-vit node-graph operator API.

Layout (only what the hot path needs - see DESIGN.md):

* ``graph`` / ``context`` / ``message`` / ``views`` - host-side mirror of the reference's
  operator API, wire codec and /compute handler (main/graph.py, main/context.py, main/message.py,
  main/views.py:30-42);
* ``nodes/cos.py`` - the reference's plumbing operator;
* ``models/vit.py`` - the ViT model plugin (in the style of static/models/vgg16.py);
* ``engine`` - ctypes binding of the C-ABI in ``include/ivit.h`` (``csrc/`` HIP kernels, gfx950);
* ``vit_config`` / ``weights`` - model variants and seeded synthetic weights.
"""
__version__ = "0.1.0"
