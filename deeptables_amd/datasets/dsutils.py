# -*- coding:utf-8 -*-
"""Bank-marketing-shaped synthetic frame standing in for `deeptables.datasets.dsutils.load_bank()` (the reference
ships the UCI CSV; no network here, and data files are outside the hot path).  Same column names and dtypes as the
UCI bank-marketing table the README example trains on (README.md:82-104), deterministic for a given seed, with a
target that depends on the features so that models learn something."""
import numpy as np
import pandas as pd

_JOBS = ['admin.', 'blue-collar', 'entrepreneur', 'housemaid', 'management', 'retired', 'self-employed', 'services',
         'student', 'technician', 'unemployed', 'unknown']
_MONTHS = ['jan', 'feb', 'mar', 'apr', 'may', 'jun', 'jul', 'aug', 'sep', 'oct', 'nov', 'dec']


def load_bank(n=5000, seed=0, missing=0.01):
    rng = np.random.default_rng(seed)
    df = pd.DataFrame({
        'id': np.arange(n),
        'age': rng.integers(18, 90, n),
        'job': rng.choice(_JOBS, n),
        'marital': rng.choice(['married', 'single', 'divorced'], n, p=[.6, .28, .12]),
        'education': rng.choice(['primary', 'secondary', 'tertiary', 'unknown'], n, p=[.15, .51, .3, .04]),
        'default': rng.choice(['no', 'yes'], n, p=[.98, .02]),
        'balance': np.round(rng.normal(1400, 3000, n)).astype(np.int64),
        'housing': rng.choice(['yes', 'no'], n, p=[.56, .44]),
        'loan': rng.choice(['no', 'yes'], n, p=[.84, .16]),
        'contact': rng.choice(['cellular', 'telephone', 'unknown'], n, p=[.65, .06, .29]),
        'day': rng.integers(1, 32, n),
        'month': rng.choice(_MONTHS, n),
        'duration': np.round(rng.gamma(2.0, 130, n)).astype(np.int64),
        'campaign': rng.integers(1, 12, n),
        'pdays': np.where(rng.random(n) < .8, -1, rng.integers(1, 400, n)),
        'previous': rng.poisson(0.6, n),
        'poutcome': rng.choice(['unknown', 'failure', 'other', 'success'], n, p=[.82, .11, .04, .03]),
    })
    score = (0.004 * df['duration'] + 1.5 * (df['poutcome'] == 'success') + 0.6 * (df['housing'] == 'no')
             + 0.4 * (df['job'].isin(['student', 'retired'])) - 0.1 * df['campaign'] + 0.00005 * df['balance'] - 2.2)
    p = 1 / (1 + np.exp(-score))
    df['y'] = np.where(rng.random(n) < p, 'yes', 'no')
    if missing > 0:                       # a few holes so the imputation step has work to do
        for c in ('job', 'balance', 'education'):
            col = df[c].astype(object)
            col[rng.random(n) < missing] = np.nan
            df[c] = col if c != 'balance' else pd.to_numeric(col)
    return df
