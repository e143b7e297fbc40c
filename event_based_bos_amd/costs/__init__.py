"""Cost plugin registry (reference: src/costs/__init__.py:3-24).

``functions`` maps ``cls.name`` -> class for every (transitive) subclass of ``CostBase`` that has been
imported when this module is executed; ``HybridCost`` is imported afterwards because it looks its
members up in ``functions``.  Besides the reference's four image-domain costs the registry holds the
two contrast costs of the contrast-maximisation loop.
isort:skip_file
"""
from .base import CostBase
from .image_domain import DifferenceNorm, FlowNorm, FlowNormPxy, ImageGradient
from .contrast import GradientMagnitude, ImageVariance


def inheritors(klass):
    """All transitive subclasses of ``klass``."""
    found, stack = set(), [klass]
    while stack:
        for child in stack.pop().__subclasses__():
            if child not in found:
                found.add(child)
                stack.append(child)
    return found


functions = {k.name: k for k in inheritors(CostBase) if hasattr(k, "name")}

# For hybrid loss
from .hybrid import HybridCost  # noqa: E402
