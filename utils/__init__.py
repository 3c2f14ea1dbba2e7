"""Drop-in import surface of the reference's `utils` package for the hot path (reference utils/{io_utils,effects,constants}.py):
the names run_kenburns_batch.py, run_segmentation.ipynb and the reference's own modules import from here, resolved to the
MI355X implementation.  Only what sits on or next to the hot path is provided; tagging, COCO export, web helpers and
the dataset tools are out of scope."""
