"""Host-side loop and conditioning glue mirroring musev.pipelines (reference musev/pipelines/pipeline_controlnet.py)."""
