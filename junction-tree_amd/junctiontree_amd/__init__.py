"""MI355X-native sum-product belief propagation behind the junctiontree API (bootstrap stub)."""
