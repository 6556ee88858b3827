from qdiff.adaptive_rounding import AdaRoundQuantizer  # noqa: F401  (identical apart from `del`s in the reference)
