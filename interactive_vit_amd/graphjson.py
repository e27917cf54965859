"""Client-format graph JSON for a chain of server nodes (schema: the browser's ``graph.js:700-758``).

Layout as the reference's ``Model.generate_graph_json`` (main/context.py:55-73) produces it - and as the
golden run through the reference pins it: node *i* sits at column ``i mod w``, row ``i div w`` of a grid
``w = floor(sqrt(n))`` wide with a 200-pixel pitch, and consecutive nodes are wired "o" -> "o".
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

PITCH = 200


def grid_position(index: int, count: int) -> Dict[str, int]:
    width = int(math.sqrt(count))
    return {"x": (index % width) * PITCH, "y": int(index / width) * PITCH}


def wire(src: int, dst: int, channel: str = "o") -> Dict:
    # the client's naming: "in_port" is the SOURCE of the edge, "out_port" its destination
    return {"in_port": {"node": src, "channel": channel}, "out_port": {"node": dst, "channel": channel}}


def chain_graph(endpoints: Sequence[str], tail: Optional[Dict] = None) -> Dict:
    """``net_node`` chain over ``endpoints``; ``tail`` (an ``instance`` dict, e.g. a ``category`` node)
    is appended, wired to the last endpoint and laid out on the grid of the lengthened chain's own size,
    the way ``static/models/vgg16.py:16-29`` appends its category node."""
    nodes: List[Dict] = []
    edges: List[Dict] = []
    for i, endpoint in enumerate(endpoints):
        nodes.append({"instance": {"kind": "net_node", "endpoint": f"{endpoint}", "params": {}},
                      "pos": grid_position(i, len(endpoints))})
        if i:
            edges.append(wire(i - 1, i))
    if tail is not None:
        i = len(nodes)
        nodes.append({"instance": tail, "pos": grid_position(i, i)})
        edges.append(wire(i - 1, i))
    return {"nodes": nodes, "edges": edges}
