"""main.py of the reference (main.py:6-36): ``python main.py --config configs/lpi/coco_lpi.json``."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from lpi_amd.retrieval.trainer import train  # noqa: E402


def main():
    args = setup_parser().parse_args()
    with open(args.config) as f:
        param = json.load(f)
    args = vars(args)
    args.update(param)
    train(args)


def setup_parser():
    parser = argparse.ArgumentParser(description='LPI continual vision-language retrieval on MI355X.')
    parser.add_argument('--config', type=str, default='./configs/lpi/coco_lpi.json', help='Json file of settings.')
    parser.add_argument('--local_rank', default=-1)
    return parser


if __name__ == '__main__':
    main()
