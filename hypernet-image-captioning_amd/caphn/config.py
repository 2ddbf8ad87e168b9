"""Process-wide switches of the module API."""

# The reference wraps every generated weight slice in a fresh nn.Parameter (utils.py:57), which
# detaches it from the hypernet graph: its hypernet never receives a gradient.
# False (default): generated weights stay attached, hypernet gradients = VJP of heads/base with
# dL/dtheta (the "intended" gradients, pinned by golden vectors).  True: literal reference behaviour.
DETACH_THETA = False
