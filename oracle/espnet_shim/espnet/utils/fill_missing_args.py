"""Stand-in for espnet.utils.fill_missing_args: back-fill defaults parsed from an empty argv."""
import argparse


def fill_missing_args(args, add_arguments):
    assert isinstance(args, argparse.Namespace) or args is None
    assert callable(add_arguments)
    default_args, _ = add_arguments(argparse.ArgumentParser()).parse_known_args([])
    args = {} if args is None else vars(args)
    for key, value in vars(default_args).items():
        if key not in args:
            args[key] = value
    return argparse.Namespace(**args)
