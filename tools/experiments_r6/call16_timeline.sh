#!/bin/bash
python tools/experiments_r6/pipeline_timeline.py default 2>&1 | tail -12
python tools/experiments_r6/pipeline_timeline.py mainstream 2>&1 | tail -12
