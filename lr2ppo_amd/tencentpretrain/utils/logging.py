"""Console (+ optional file) logger in the reference's format (tencentpretrain/utils/logging.py:4-19), so
training logs diff cleanly against logs/ppo_logs/* of the reference."""
import logging


def init_logger(args):
    fmt = logging.Formatter("[%(asctime)s %(levelname)s] %(message)s")
    logger = logging.getLogger()
    logger.setLevel(getattr(args, "log_level", "INFO"))
    console = logging.StreamHandler()
    console.setFormatter(fmt)
    logger.handlers = [console]
    if getattr(args, "log_path", None) is not None:
        fh = logging.FileHandler(args.log_path, encoding="UTF-8")
        fh.setLevel(getattr(args, "log_file_level", "INFO"))
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger
