"""Attribute keys and node-name conventions of the problem graphs.

The values are part of the drop-in surface: code written against GraphIK indexes graphs with
`graph[u][v][DIST]`, `graph.nodes[n][POS]` and so on, so the names exported here and the strings
behind them have to be the ones GraphIK uses (graphik/utils/constants.py).  Here they are grouped
by what the array-backed graphs of this package store under them.
"""

# -- node naming: "p<i>" is the origin of joint i, "q<i>" the auxiliary point one axis_length
#    along its rotation axis; the chain is rooted at ROOT -------------------------------------
MAIN_PREFIX, AUX_PREFIX = "p", "q"
ROOT = MAIN_PREFIX + "0"
BASE = "base"

# -- per-node attributes ----------------------------------------------------------------------
POS = "pos"              # point coordinates (k,)
TYPE = "type"            # list of ROBOT / OBSTACLE / END_EFFECTOR / BASE tags
RADIUS = "radius"        # spherical obstacles only
ROBOT, OBSTACLE, END_EFFECTOR = "robot", "obstacle", "end_effector"

# -- per-edge attributes ----------------------------------------------------------------------
DIST = "weight"          # known Euclidean distance (networkx's default weight key)
LOWER, UPPER = "lower_limit", "upper_limit"   # distance bounds from joint limits / obstacles
BOUNDED = "bounded"      # which of the two bounds is informative: list of BELOW / ABOVE
BELOW, ABOVE = "below", "above"
TRANSFORM = "T"          # homogeneous transform between the two frames of a robot-graph edge

UNDEFINED = None
