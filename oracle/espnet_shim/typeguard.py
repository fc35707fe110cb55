"""Stand-in for `typeguard` (imported by reference variance_predictor.py:8; the call is commented out at :46)."""


def check_argument_types(*args, **kwargs):
    return True
