"""Bayes-by-backprop layers of the meta-regularised (MR) models (reference: networks/bbb/)."""
