"""Observation ingestion: from the reference's input formats to the
batched NaN-encoded ``[R,T,N]`` buffer the kernels read.

Host side (pandas), mirroring what ``Metran.__init__`` does to its ``oseries`` argument for ONE model:

* ``read_series_csv``       the example format ``date,residuals`` (/root/reference/examples/data/*_res.csv,
                            read in examples/metran_practical_example.ipynb with ``read_csv(index_col=0,
                            parse_dates=True)``)
* ``combine_series``        ``Metran.set_observations`` (metran/metran.py:508-579): list/tuple/DataFrame ->
                            one DataFrame, ``truncate`` (:123-148: tmin/tmax, drop all-NaN dates), daily grid
                            (``asfreq("D")``, :571); same exceptions and messages
* ``cross_section_pairs``   ``Metran.test_cross_section`` (metran/metran.py:150-199)
* ``standardize``           ``Metran.standardize`` (metran/metran.py:102-121), host version

and for MANY models:

* ``ObservationBatch``      stacks the frames of R models into one ``[R,T,N]`` array (NaN = missing; shorter
                            records are padded with all-NaN steps at the END, which the filter skips:
                            ``observation_count == 0``, kalmanfilter.py:336) and uploads it ONCE;
                            standardisation then runs on the device (``BatchedKalman.standardize`` ->
                            ``mk_standardize``), masks are applied there (``mask_observations`` ->
                            ``mk_mask_observations``) and un-masking costs nothing.
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)

__all__ = ["read_series_csv", "combine_series", "cross_section_pairs", "standardize", "ObservationBatch"]


def read_series_csv(path, name=None):
    """One series from a ``date,value`` CSV file -> ``pandas.Series`` with a DatetimeIndex."""
    import os

    from pandas import read_csv

    frame = read_csv(path, index_col=0, parse_dates=True)
    series = frame.iloc[:, 0]
    series.name = name if name is not None else os.path.splitext(os.path.basename(str(path)))[0]
    return series


def combine_series(oseries, tmin=None, tmax=None):
    """``Metran.set_observations`` + ``truncate``: returns ``(frame, names)`` with ``frame`` on a daily grid
    (NaN where a series has no observation)."""
    from pandas import DataFrame, DatetimeIndex, Series, concat

    if isinstance(oseries, (list, tuple)):
        _oseries, names = [], []
        if len(oseries) > 1:
            for i, os_ in enumerate(oseries):
                if hasattr(os_, "series") and hasattr(os_, "name") and not isinstance(os_, (Series, DataFrame)):
                    _oseries.append(os_.series)  # pastas.TimeSeries duck type (metran.py:539-541)
                    names.append(os_.name)
                elif isinstance(os_, (Series, DataFrame)):
                    if isinstance(os_, DataFrame):
                        if os_.shape[1] > 1:
                            msg = "One or more series have DataFrame with multiple columns"
                            logger.error(msg)
                            raise Exception(msg)
                        os_ = os_.squeeze()
                    if os_.name is None:
                        os_.name = "Series" + str(i + 1)
                    _oseries.append(os_)
                    names.append(os_.name)
            frame = concat(_oseries, axis=1)
        else:
            frame = DataFrame()
    elif isinstance(oseries, DataFrame):
        frame = oseries
        names = list(oseries.columns)
    else:
        msg = "Input type should be either a list, tuple, or pandas.DataFrame"
        logger.error(msg)
        raise TypeError(msg)
    if frame.shape[1] < 2:
        msg = "Metran requires at least 2 series, found " + str(frame.shape[1])
        logger.error(msg)
        raise Exception(msg)
    lo = frame.index.min() if tmin is None else tmin
    hi = frame.index.max() if tmax is None else tmax
    frame = frame.loc[lo:hi].dropna(how="all")
    if not isinstance(frame.index, DatetimeIndex):
        msg = "Index of series must be DatetimeIndex"
        logger.error(msg)
        raise TypeError(msg)
    return frame.asfreq("D"), list(names)


def cross_section_pairs(frame, min_pairs=20):
    """``Metran.test_cross_section``: for each series the number of dates at which it is observed (that is
    what the reference's ``dropna(subset=[s])["count"].count()`` evaluates to); raises like the reference
    when a series has fewer than ``max(min_pairs, 1)``."""
    if min_pairs == 0:
        logger.warning("min_pairs must be greater than 0.")
    pairs = frame.count(axis=0)
    if pairs.min() < max(min_pairs, 1):
        err = pairs[pairs < min_pairs].index.tolist()
        msg = "Number of cross-sectional data is less than " + str(min_pairs) + " for series " + (", ").join(
            [str(e) for e in err])
        logger.error(msg)
        raise Exception(msg)
    return pairs


def standardize(frame):
    """``Metran.standardize`` on the host: ``(standardised frame, std, mean)``."""
    std = frame.std()
    mean = frame.mean()
    return (frame - mean) / std, np.array(std.values), np.array(mean.values)


class ObservationBatch:
    """R models' observation frames stacked for the device.

    Parameters
    ----------
    models : sequence
        one entry per model, each whatever ``Metran(oseries)`` accepts (list/tuple of Series, DataFrame);
        all models must have the same number of series N.
    tmin, tmax, min_pairs : as in ``Metran.settings``

    Attributes
    ----------
    obs : float64 ``[R,T,N]``   raw (un-standardised) values, NaN = missing, T = longest record
    lengths : int64 ``[R]``     number of daily steps of each record (the rest is NaN padding)
    names : list of lists       series names per model
    index : list                DatetimeIndex per model
    """

    def __init__(self, models, tmin=None, tmax=None, min_pairs=20):
        frames, self.names, self.index = [], [], []
        for m in models:
            frame, names = combine_series(m, tmin=tmin, tmax=tmax)
            cross_section_pairs(frame, min_pairs=min_pairs)
            frames.append(frame)
            self.names.append(names)
            self.index.append(frame.index)
        if not frames:
            raise ValueError("no models")
        N = frames[0].shape[1]
        if any(f.shape[1] != N for f in frames):
            raise ValueError("all models of a batch must have the same number of series")
        self.lengths = np.array([f.shape[0] for f in frames], dtype=np.int64)
        T = int(self.lengths.max())
        self.obs = np.full((len(frames), T, N), np.nan)
        for r, f in enumerate(frames):
            self.obs[r, : f.shape[0]] = f.values
        self.mean = self.std = None

    @property
    def shape(self):
        return self.obs.shape

    def upload(self, kf):
        """Upload once, standardise on the device; ``kf`` then holds the standardised records and the
        scaling that brings projections back to the original units.  Returns ``kf``."""
        mean, std = kf.standardize(self.obs)
        self.mean, self.std = mean.cpu().numpy(), std.cpu().numpy()
        return kf

    def frame(self, r, values):
        """Wrap ``values [T,N]`` (e.g. ``sim_means[r]``) of model r back into a DataFrame on its own index."""
        from pandas import DataFrame

        return DataFrame(np.asarray(values)[: self.lengths[r]], index=self.index[r], columns=self.names[r])
