from .maskformer_criterion import SetCriterion  # noqa: F401
from .matcher import Point_HungarianMatcher  # noqa: F401
