"""Stand-in for espnet.utils.cli_utils.strtobool."""


def strtobool(x):
    s = str(x).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError("invalid truth value %r" % (x,))
