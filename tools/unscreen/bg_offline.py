"""The reference names this mode both `bg_step` (README) and `bg_offline` (file name): same entry point."""
import runpy
import os.path as osp

if __name__ == '__main__':
    runpy.run_path(osp.join(osp.dirname(osp.abspath(__file__)), "bg_step.py"), run_name="__main__")
