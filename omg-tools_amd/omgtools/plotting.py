"""Plot hooks kept as no-ops.

The reference mixes a matplotlib/Tk `PlotLayer` into every object
(`execution/plotlayer.py:180-405`, forcing the TkAgg backend at import, line
25).  Visualisation is outside the hot-path scope; the
methods exist so reference scripts that call `problem.plot('scene')` or
`vehicle.plot('input', ...)` run headless unchanged.
"""


class PlotLayer(object):
    simulator = None

    def __init__(self):
        self.plots = []

    def plot(self, *args, **kwargs):
        return None

    def update_plots(self):
        return None

    def save_plot(self, *args, **kwargs):
        return None

    def plot_movie(self, *args, **kwargs):
        return None

    def save_movie(self, *args, **kwargs):
        return None
