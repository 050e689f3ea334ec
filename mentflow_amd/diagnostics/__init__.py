from .diagnostics import Diagnostic, Histogram, Histogram1D, Histogram2D, Projection
