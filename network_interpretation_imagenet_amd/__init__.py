"""MI355X-native masked-perturbation saliency engine."""
